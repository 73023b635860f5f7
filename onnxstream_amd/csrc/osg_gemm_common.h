// Shared by the contraction kernels of libosgpu (osg_gemm.hip, osg_conv3x3.hip): launch parameters and the fused epilogue.
#pragma once
#include <utility>
#include <vector>
#include "osg_common.h"

namespace osg_mm {

struct GemmParams {
    const f16* A;
    const f16* Bt;
    f16* C;
    const void* bias;
    const f16* residual;
    const f16* rowbias;          // optional [M / rb_rows][rb_ld] per-image channel bias (the resnet time-embedding add)
    int rb_rows;
    long rb_ld;
    float* partial;
    int M, N, K;
    long lda;
    long strideA, strideB, strideC;
    int bias_f32, act;
    int splits, k_per_split;
    // conv geometry (CONV only)
    int H, W, Cin, Ho, Wo, KW, sh, sw, pt, pl;
    // v2 (direct-to-LDS) kernel only
    unsigned a_bytes, b_bytes;   // buffer-descriptor extents of one batch item of A / Bt
    long a_bytes_l;              // conv: byte size of the whole NHWC input (host side, before the 2 GiB check)
    int mt, nt, n_major;         // tile grid and the order tiles are walked inside an XCD's contiguous chunk
    int* tickets;                // split-K arrival counters (one per output tile), zero between launches
    const float* pre_tab;        // osg_conv3x3.hip PRE variant: per-image affine table [n][2][Cin] (ca, cb) of a fused GroupNorm on the INPUT
    int pre_act, pre_imgs;       //   activation applied after the affine (OSG_ACT_SILU), number of images
    float w_scale;               // W8 kernels (osg_gemm_w8.hip): Bt holds uint8 codes, w = (q - w_zp) * w_scale
    int w_zp;
    const float* ln_c1;          // osg_gemm_ln: LayerNorm over K folded into this GEMM -- c1[n] = sum_k W'[n][k]; bias holds c2 (f32)
    float ln_eps;
    const float* rs_in;          //   row statistics of A emitted by the GEMM that produced it: [M][K/32][2] (sum, sum of squares) per 32 columns
    float* rs_out;               // osg_gemm_rowstats: this GEMM's epilogue also emits [M][rs_np][2] partial row statistics of its f16 output
    int rs_np;                   //   = N / 32
};

// ---- LayerNorm folded into the consuming GEMM (osg_gemm_ln) ----------------------------------------------------------------
// The math waves already read every A fragment of their rows out of LDS for the MFMAs: two v_dot2c_f32_f16 per f16 pair accumulate
// the row's sum and sum of squares on the side (fp32, in the shadow of the MFMA pipe).  A lane holds row (lane & 15) + 16 i and the
// k-chunk (lane >> 4) of every 32-deep half, so the row totals are a 2-step butterfly over the four 16-lane groups -- and they are
// the rows of this lane's own accumulators: no LDS, no extra pass over A.
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
template <int TM>
__device__ __forceinline__ void ln_accumulate(const f16x8 (&a)[TM], float (&ls)[TM], float (&lq)[TM]) {
    const f16x2v ones = {(f16)1.0f, (f16)1.0f};
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const f16x2v pr = {a[i][2 * e], a[i][2 * e + 1]};
            ls[i] = __builtin_amdgcn_fdot2(pr, ones, ls[i], false);
            lq[i] = __builtin_amdgcn_fdot2(pr, pr, lq[i], false);
        }
}
// acc <- rstd_m * (acc - mean_m * c1[n]): what the GEMM of the NORMALISED rows with the gamma-folded weight would have accumulated
template <int TM, int TN, bool PARTIAL>
__device__ __forceinline__ void ln_apply(const GemmParams& p, f32x4 (&acc)[TM][TN], float (&ls)[TM], float (&lq)[TM], int n0, int wn0, int lane) {
#pragma unroll
    for (int i = 0; i < TM; i++) {
        float s = ls[i], q = lq[i];
        if (PARTIAL) {   // the four 16-lane groups hold different k-chunks of the row
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
        }
        // E[x^2] - mean^2 in f64 (two fma-rate operations; the f32 subtraction would cancel), everything else in f32: an f64 divide
        // or square root costs hundreds of instructions per row here
        const double ik = (double)(1.0f / (float)p.K);
        const double mean_d = (double)s * ik;
        const float var = fmaxf((float)((double)q * ik - mean_d * mean_d), 0.f);
        const float mean = (float)mean_d;
        const float rstd = 1.0f / sqrtf(var + p.ln_eps);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) c = *reinterpret_cast<const f32x4*>(p.ln_c1 + n);
#pragma unroll
            for (int r = 0; r < 4; r++) acc[i][j][r] = rstd * (acc[i][j][r] - mean * c[r]);
        }
    }
}

// ---- epilogue shared by both kernels: lane owns C[m][n..n+3], m = tile_m + (lane&15), n = tile_n + (lane>>4)*4 --------
// (operands are swapped -- weights feed the MFMA "A" port -- so the 4 accumulator registers of a lane are 4 consecutive
// output channels of one pixel); bias/residual/activation fused in f32 before the single RNE rounding to f16.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane,
                                              int zb, int zslab) {
    const int N = p.N;
    if (p.splits == 1) {
        f16* __restrict__ C = p.C + zb * p.strideC;
        const f16* __restrict__ R = p.residual ? p.residual + zb * p.strideC : nullptr;
        const bool vec_ok = (N & 3) == 0;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = m0 + wm0 + i * 16 + (lane & 15);
            if (m >= p.M) continue;
            float rs_s[(TN + 1) / 2], rs_q[(TN + 1) / 2];   // rs_out: sums over this wave's 32-column slots of row m (of the ROUNDED outputs)
#pragma unroll
            for (int h = 0; h < (TN + 1) / 2; h++) rs_s[h] = rs_q[h] = 0.f;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
                if (n >= N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (vec_ok) {
                    if (p.bias) {
                        if (p.bias_f32) {
                            f32x4 bv = *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
#pragma unroll
                            for (int r = 0; r < 4; r++) v[r] += bv[r];
                        } else {
                            f16x4 bv = *reinterpret_cast<const f16x4*>((const f16*)p.bias + n);
#pragma unroll
                            for (int r = 0; r < 4; r++) v[r] += (float)bv[r];
                        }
                    }
                    if (p.rowbias) {
                        f16x4 rb = *reinterpret_cast<const f16x4*>(p.rowbias + (long)(m / p.rb_rows) * p.rb_ld + n);
#pragma unroll
                        for (int r = 0; r < 4; r++) v[r] += (float)rb[r];
                    }
                    if (R) {
                        f16x4 rv = *reinterpret_cast<const f16x4*>(R + (long)m * N + n);
#pragma unroll
                        for (int r = 0; r < 4; r++) v[r] += (float)rv[r];
                    }
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (f16)osg_apply_act(v[r], p.act);
                    *reinterpret_cast<f16x4*>(C + (long)m * N + n) = o;
                    if (p.rs_out) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float f = (float)o[r];
                            rs_s[j >> 1] += f;
                            rs_q[j >> 1] = fmaf(f, f, rs_q[j >> 1]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (n + r >= N) break;
                        float x = v[r];
                        if (p.bias) x += p.bias_f32 ? ((const float*)p.bias)[n + r] : (float)((const f16*)p.bias)[n + r];
                        if (p.rowbias) x += (float)p.rowbias[(long)(m / p.rb_rows) * p.rb_ld + n + r];
                        if (R) x += (float)R[(long)m * N + n + r];
                        C[(long)m * N + n + r] = (f16)osg_apply_act(x, p.act);
                    }
                }
            }
            if (p.rs_out) {
                // the four 16-lane groups of a row hold different columns of the same slot pair: 2-step butterfly, one lane group stores
#pragma unroll
                for (int h = 0; h < TN / 2; h++) {
                    float s = rs_s[h], q = rs_q[h];
                    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
                    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                    const int slot = ((n0 + wn0) >> 5) + h;
                    if ((lane >> 4) == 0 && slot < p.rs_np) {
                        float* d = p.rs_out + ((long)m * p.rs_np + slot) * 2;
                        d[0] = s;
                        d[1] = q;
                    }
                }
            }
        }
    } else {
        float* __restrict__ P = p.partial + ((long)zslab) * p.M * N;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = m0 + wm0 + i * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
                if (n >= N) continue;
                if ((N & 3) == 0) {
                    if (p.tickets) {
                        // write-through (sc1): the slab goes past this XCD's L2 to memory, so the last-arriving block of the tile -- maybe on
                        // another XCD -- can read it with sc1 loads after the ticket, no release / acquire fence on either side
                        const float* dst = P + (long)m * N + n;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(acc[i][j]) : "memory");
                    } else
                        *reinterpret_cast<f32x4*>(P + (long)m * N + n) = acc[i][j];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (n + r < N) P[(long)m * N + n + r] = acc[i][j][r];
                }
            }
        }
    }
}


// ---- fused GEGLU epilogue (act == OSG_ACT_GEGLU): the weight rows were pair-interleaved in blocks of 16 at plan time, so MFMA
// tile 2j holds 16 "value" columns and tile 2j+1 the matching 16 "gate" columns; out[m][c] = (v + bv) * gelu_erf(g + bg), written
// to a [M, N/2] matrix.  Replaces Add(bias) + Slice,Slice,Div,Erf,Add,Mul,Mul,Mul (reference src/onnxstream.cpp:6499,4001,...).
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_geglu(const GemmParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane,
                                                    int zb) {
    static_assert(TN % 2 == 0, "GEGLU epilogue needs value/gate tile pairs");
    const int No = p.N >> 1;
    f16* __restrict__ C = p.C + zb * p.strideC;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = m0 + wm0 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
            const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;      // column of the value tile in the interleaved space
            if (n >= p.N) continue;
            const int c = ((n0 + wn0 + j * 16) >> 1) + (lane >> 4) * 4;   // output column
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = acc[i][j][r], g = acc[i][j + 1][r];
                if (p.bias) {
                    v += p.bias_f32 ? ((const float*)p.bias)[n + r] : (float)((const f16*)p.bias)[n + r];
                    g += p.bias_f32 ? ((const float*)p.bias)[n + 16 + r] : (float)((const f16*)p.bias)[n + 16 + r];
                }
                o[r] = (f16)(v * osg_gelu_erf(g));
            }
            *reinterpret_cast<f16x4*>(C + (long)m * No + c) = o;
        }
    }
}

// ---- split-K without a reduce launch: the LAST k-slice block to arrive at an output tile folds the f32 slabs (in slab order => the same
// bits whoever arrives last) and writes the f16 tile.  Slabs are published write-through (sc1 stores, see gemm_epilogue), every wave drains
// its stores (vmcnt(0)) before the block's single relaxed agent-scope ticket; the reducer reads the slabs with sc1 loads (they bypass its
// L1 and revalidate against memory): the hand-off needs neither the release fence (an L2 write-back of freshly dirtied slabs, ~6 us per
// block) nor the acquire.  Call with every thread of the block that ran gemm_epilogue (nthr of them, tid = 0 .. nthr-1).
template <int BM, int BN>
__device__ __forceinline__ void splitk_finish(const GemmParams& p, int m0, int n0, int tile_id, int zb, int* flag, int tid, int nthr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(&p.tickets[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag != p.splits - 1) return;
    if (tid == 0) __hip_atomic_store(&p.tickets[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    constexpr int VPR = BN / 4;
    const long MN = (long)p.M * p.N;
    const float* __restrict__ P0 = p.partial + (long)zb * p.splits * MN;
    f16* __restrict__ C = p.C + zb * p.strideC;
    const f16* __restrict__ R = p.residual ? p.residual + zb * p.strideC : nullptr;
    for (int v = tid; v < BM * VPR; v += nthr) {
        const int r = v / VPR, m = m0 + r, n = n0 + (v - r * VPR) * 4;
        if (m >= p.M || n >= p.N) continue;
        const float* src = P0 + (long)m * p.N + n;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < p.splits; s0 += 4) {
            f32x4 part[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float* a = src + (long)min(s0 + u, p.splits - 1) * MN;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[u]) : "v"(a) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; u++) {
                asm volatile("" : "+v"(part[u]));
                if (s0 + u < p.splits) sum += part[u];
            }
        }
        if (p.bias) {
            if (p.bias_f32) sum += *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
            else {
                f16x4 bv = *reinterpret_cast<const f16x4*>((const f16*)p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; e++) sum[e] += (float)bv[e];
            }
        }
        if (p.rowbias) {
            f16x4 rb = *reinterpret_cast<const f16x4*>(p.rowbias + (long)(m / p.rb_rows) * p.rb_ld + n);
#pragma unroll
            for (int e = 0; e < 4; e++) sum[e] += (float)rb[e];
        }
        if (R) {
            f16x4 rv = *reinterpret_cast<const f16x4*>(R + (long)m * p.N + n);
#pragma unroll
            for (int e = 0; e < 4; e++) sum[e] += (float)rv[e];
        }
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (f16)osg_apply_act(sum[e], p.act);
        *reinterpret_cast<f16x4*>(C + (long)m * p.N + n) = o;
    }
}

int launch_splitk_reduce(osg_ctx* ctx, const GemmParams& p, int batch);   // osg_gemm.hip

}  // namespace osg_mm

// osg_conv3x3.hip: halo-reuse 3x3 / stride 1 / pad 1 convolution.  Returns -1 when the shape is not one it takes.
int osg_conv3x3_run(osg_ctx* ctx, osg_mm::GemmParams& p);
// the same in pieces, for the measured configuration choice (osg_tune.h): shape gate, ranked (BN, splits) candidates, one launch
int osg_conv3x3_prepare(osg_ctx* ctx, osg_mm::GemmParams& p);
std::vector<std::pair<double, std::pair<int, int>>> osg_conv3x3_rank(const osg_ctx* ctx, const osg_mm::GemmParams& p);
int osg_conv3x3_launch(osg_ctx* ctx, osg_mm::GemmParams p, int bn, int splits);
int osg_conv3x3_supported(int N, int H, int W, int Cin, int Cout);
