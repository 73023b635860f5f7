// libosgpu: dense contractions on the gfx950 matrix cores.
//
// One implicit-GEMM kernel serves Conv (XnnPack::convolution, reference onnxstream.cpp:1292) and MatMul/Gemm
// (XnnPack::matrix_multiply, :1035):   C[M,N] = A[M,K] * Bt[N,K]^T   (f16 operands, f32 accumulate on
// v_mfma_f32_16x16x32_f16, one RNE rounding to f16 in the epilogue -- the numerics class of XNNPACK's f16_f32acc GEMMs).
//   * A is either a plain row-major matrix or gathered on the fly from an NHWC activation (im2col never materialised):
//     m -> (n,ho,wo), k -> (kh,kw,c); a 16-byte k-chunk never straddles a filter tap when Cin % 8 == 0.
//   * Bt is K-contiguous: OHWI conv weights already are [Cout][KH*KW*Cin]; MatMul weights ([K,N] on disk) are
//     re-laid out to [N,K] once when they become resident.
//   * global -> registers -> LDS (padded rows: +16 B => conflict-free ds_read_b128), double-buffered, one barrier per
//     k-tile, next tile's global loads in flight under the MFMAs.
//   * operands are swapped (weights as MFMA "A") so each lane owns 4 consecutive output channels of one pixel:
//     8-byte epilogue stores, bias/residual fused in f32 before the single rounding.
//   * split-K (grid.z) with f32 partial slabs + a fused reduce epilogue for the small-M (8x8, 16x16 latent) layers.
#include "osg_gemm_common.h"
#include "osg_tune.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace osg_mm;

namespace {

template <int BM, int BN, int BK, bool CONV, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    constexpr int LDS = BK + 8;               // padded row stride (halves)
    constexpr int CPR = BK / 8;               // 16-byte chunks per tile row
    constexpr int ROWS_PER_PASS = 256 / CPR;  // rows covered by one pass of the 256 threads
    constexpr int A_IT = BM / ROWS_PER_PASS;
    constexpr int B_IT = BN / ROWS_PER_PASS;
    constexpr int WM = BM / 2, WN = BN / 2;   // 2x2 waves
    constexpr int TM = WM / 16, TN = WN / 16;
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for 256 threads");

    extern __shared__ __attribute__((aligned(16))) f16 smem[];
    f16* As0 = smem;
    f16* Bs0 = smem + BM * LDS;
    constexpr int STAGE = (BM + BN) * LDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;

    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int zb = blockIdx.z / p.splits;
    const int zs = blockIdx.z - zb * p.splits;
    const int kbeg = zs * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const f16* __restrict__ A = p.A + zb * p.strideA;
    const f16* __restrict__ Bt = p.Bt + zb * p.strideB;

    const int kc = tid % CPR;       // this thread's chunk column (same for all its chunks)
    const int r0 = tid / CPR;

    // per-chunk row state
    long a_off[A_IT];
    int a_hi0[A_IT], a_wi0[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; i++) {
        int m = m0 + r0 + i * ROWS_PER_PASS;
        a_ok[i] = m < p.M;
        if (CONV) {
            int mm = a_ok[i] ? m : 0;
            int hw = p.Ho * p.Wo;
            int n_img = mm / hw;
            int rem = mm - n_img * hw;
            int ho = rem / p.Wo;
            int wo = rem - ho * p.Wo;
            a_off[i] = (long)n_img * p.H * p.W * p.Cin;
            a_hi0[i] = ho * p.sh - p.pt;
            a_wi0[i] = wo * p.sw - p.pl;
        } else {
            a_off[i] = (long)(a_ok[i] ? m : 0) * p.lda;
            a_hi0[i] = a_wi0[i] = 0;
        }
    }
    long b_off[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; i++) {
        int n = n0 + r0 + i * ROWS_PER_PASS;
        b_ok[i] = n < p.N;
        b_off[i] = (long)(b_ok[i] ? n : 0) * p.K;
    }

    f16x8 areg[A_IT], breg[B_IT];

    auto load_tile = [&](int k0) {
        const int k = k0 + kc * 8;
        if (VEC) {
            const bool kok = k < kend;
            int kh = 0, kw = 0, c = k;
            if (CONV) {
                int cell = k / p.Cin;
                c = k - cell * p.Cin;
                kh = cell / p.KW;
                kw = cell - kh * p.KW;
            }
#pragma unroll
            for (int i = 0; i < A_IT; i++) {
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (CONV) {
                    int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                    if (a_ok[i] && kok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                        v = *reinterpret_cast<const f16x8*>(A + a_off[i] + ((long)hi * p.W + wi) * p.Cin + c);
                } else {
                    if (a_ok[i] && kok) v = *reinterpret_cast<const f16x8*>(A + a_off[i] + k);
                }
                areg[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_IT; i++) {
                f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (b_ok[i] && kok) v = *reinterpret_cast<const f16x8*>(Bt + b_off[i] + k);
                breg[i] = v;
            }
        } else {
            // generic path: K or Cin not a multiple of 8 (e.g. conv_in with 4 input channels) -- element-wise gather
#pragma unroll
            for (int i = 0; i < A_IT; i++) {
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    int ke = k + e;
                    f16 x = (f16)0;
                    if (a_ok[i] && ke < kend) {
                        if (CONV) {
                            int cell = ke / p.Cin;
                            int c = ke - cell * p.Cin;
                            int kh = cell / p.KW;
                            int kw = cell - kh * p.KW;
                            int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                                x = A[a_off[i] + ((long)hi * p.W + wi) * p.Cin + c];
                        } else {
                            x = A[a_off[i] + ke];
                        }
                    }
                    v[e] = x;
                }
                areg[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_IT; i++) {
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    int ke = k + e;
                    v[e] = (b_ok[i] && ke < kend) ? Bt[b_off[i] + ke] : (f16)0;
                }
                breg[i] = v;
            }
        }
    };

    auto store_tile = [&](int buf) {
        f16* As = As0 + buf * STAGE;
        f16* Bs = Bs0 + buf * STAGE;
#pragma unroll
        for (int i = 0; i < A_IT; i++)
            *reinterpret_cast<f16x8*>(As + (r0 + i * ROWS_PER_PASS) * LDS + kc * 8) = areg[i];
#pragma unroll
        for (int i = 0; i < B_IT; i++)
            *reinterpret_cast<f16x8*>(Bs + (r0 + i * ROWS_PER_PASS) * LDS + kc * 8) = breg[i];
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (nkt > 0) {
        load_tile(kbeg);
        store_tile(0);
    }
    __syncthreads();

    const int frow = lane & 15;
    const int fk = (lane >> 4) * 8;
    for (int kt = 0; kt < nkt; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kbeg + (kt + 1) * BK);
        const f16* As = As0 + cur * STAGE;
        const f16* Bs = Bs0 + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ks++) {
            f16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++)
                a[i] = *reinterpret_cast<const f16x8*>(As + (wm0 + i * 16 + frow) * LDS + ks * 32 + fk);
#pragma unroll
            for (int j = 0; j < TN; j++)
                b[j] = *reinterpret_cast<const f16x8*>(Bs + (wn0 + j * 16 + frow) * LDS + ks * 32 + fk);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }

    gemm_epilogue<TM, TN>(p, acc, m0, n0, wm0, wn0, lane, zb, (int)blockIdx.z);
}

}  // namespace
#include "osg_gemm2.h"
namespace {

// The SD / SDXL conv_in and the VAE decoder's first convolution (3x3, Cin = 4: K = 36) on the matrix cores: the vector kernel this replaced (round 4) spent 288 FMAs + 288
// f16 -> f32 conversions per (pixel, 8 channels) and took 31 us for 0.2 GFLOP at 2 x 64 x 64 -> 320 (profiles/r04_breakdown_tail_v4_fastbox.txt), a
// twentieth of a pass's convolution time for a ten-thousandth of its work.  Here K is padded to 64 (two v_mfma_f32_16x16x32_f16 steps, the second one holds tap
// 8 and zeros): the filter bank sits in LDS as [Cout][64 + 8] with the padding zeroed, a lane builds its A fragment -- pixel l & 15, taps 2g and 2g + 1 -- from
// three 8-byte loads, and each wave walks every other 16-channel tile of its 16 pixels.  Bias, per-image bias, residual, activation and the second destination in the epilogue.
__global__ __launch_bounds__(256) void conv_cin4_mfma_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char cin4_lds[];
    constexpr int WLD = 144;                         // bytes per filter row in LDS: 64 halves + 8 of padding (conflict-free 16-byte fragment reads)
    const int N = p.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    float* const bsm = reinterpret_cast<float*>(cin4_lds + (size_t)N * WLD);          // bias as f32 [N]
    for (int i = tid; i < N * 8; i += 256) {         // (row n, 16-byte chunk c): k = 8c .. 8c + 7 of a 36-deep row, rows 72 bytes apart in memory
        const int n = i >> 3, c = i & 7;
        f16x4 lo = f16x4{0, 0, 0, 0}, hi = f16x4{0, 0, 0, 0};
        const f16* src = p.Bt + (long)n * 36 + c * 8;
        if (c < 5) lo = *reinterpret_cast<const f16x4*>(src);
        if (c < 4) hi = *reinterpret_cast<const f16x4*>(src + 4);
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        *reinterpret_cast<f16x8*>(cin4_lds + n * WLD + c * 16) = v;
    }
    for (int n = tid; n < N; n += 256) bsm[n] = !p.bias ? 0.f : p.bias_f32 ? ((const float*)p.bias)[n] : (float)((const f16*)p.bias)[n];
    // this lane's pixel and its A fragments
    const int m = blockIdx.x * 32 + (wave & 1) * 16 + l16;
    const bool live = m < p.M;
    const int mc = live ? m : p.M - 1;
    const int hw = p.Ho * p.Wo;
    const int n_img = mc / hw, r2 = mc - n_img * hw, ho = r2 / p.Wo, wo = r2 - ho * p.Wo;
    const f16* xin = p.A + (long)n_img * p.H * p.W * 4;
    auto tap = [&](int t) __attribute__((always_inline)) {
        const int hi = ho * p.sh - p.pt + t / 3, wi = wo * p.sw - p.pl + t % 3;
        const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const f16x4 v = *reinterpret_cast<const f16x4*>(xin + ((long)(ok ? hi : 0) * p.W + (ok ? wi : 0)) * 4);
        return ok ? v : f16x4{0, 0, 0, 0};
    };
    const f16x4 t0 = tap(2 * g), t1 = tap(2 * g + 1), t8 = tap(8);
    f16x8 a0, a1;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        a0[e] = t0[e]; a0[4 + e] = t1[e];
        a1[e] = g == 0 ? t8[e] : (f16)0; a1[4 + e] = (f16)0;
    }
    __syncthreads();
    const long ldc = p.ldc ? p.ldc : (long)N;
    const int img_row = m / p.rb_rows;
    for (int j = wave >> 1; j < N / 16; j += 2) {
        const char* wrow = cin4_lds + (j * 16 + l16) * WLD + g * 16;
        const f16x8 b0 = *reinterpret_cast<const f16x8*>(wrow), b1 = *reinterpret_cast<const f16x8*>(wrow + 64);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, a1, acc, 0, 0, 0);
        if (!live) continue;
        const int n = j * 16 + g * 4;                // the lane holds channels n .. n + 3 of pixel m
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bsm + n);
        f16x4 rb = f16x4{0, 0, 0, 0}, rs = f16x4{0, 0, 0, 0};
        if (p.rowbias) rb = *reinterpret_cast<const f16x4*>(p.rowbias + (long)img_row * p.rb_ld + n);
        if (p.residual) rs = *reinterpret_cast<const f16x4*>(p.residual + (long)m * N + n);
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = acc[r] + bv[r];
            if (p.rowbias) v += (float)rb[r];
            if (p.residual) v += (float)rs[r];
            o[r] = (f16)osg_apply_act(v, p.act);
        }
        *reinterpret_cast<f16x4*>(p.C + (long)m * ldc + n) = o;
        if (p.C2) *reinterpret_cast<f16x4*>(p.C2 + (long)m * p.ldc2 + n) = o;
    }
}

// sum the split-K slabs, fuse bias/residual/activation, round once to f16
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, f16* __restrict__ C,
                                                            const void* __restrict__ bias, int bias_f32,
                                                            const f16* __restrict__ residual, long MN, int N, int splits, int batch,
                                                            long strideC, int act, const f16* __restrict__ rowbias, int rb_rows, long rb_ld, long ldc,
                                                            f16* __restrict__ C2, long ldc2) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    long total = MN * batch;
    if (idx >= total) return;
    int b = (int)(idx / MN);
    long e = idx - (long)b * MN;
    float v = 0.f;
    for (int s = 0; s < splits; s++) v += partial[((long)(b * splits + s)) * MN + e];
    int n = (int)(e % N);
    if (bias) v += bias_f32 ? ((const float*)bias)[n] : (float)((const f16*)bias)[n];
    if (rowbias) v += (float)rowbias[(e / N / rb_rows) * rb_ld + n];
    if (residual) v += (float)residual[b * strideC + e];
    const f16 o = (f16)osg_apply_act(v, act);
    if (ldc == N && !C2) C[b * strideC + e] = o;
    else {   // output views (batch == 1): rows ldc apart, optional second copy
        const long m = e / N;
        C[m * ldc + n] = o;
        if (C2) C2[m * ldc2 + n] = o;
    }
}

// ... the same for N % 4 == 0 (every layer of the UNet), four outputs per thread: 16-byte slab loads, ALL of a thread's loads (up to 8 slices at a time,
// the epilogue operands first) in flight before the first addition.  Per element the slices are added in the same order (0, 1, 2, ...) as above: same bits.
template <int SB>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ partial, f16* __restrict__ C, const void* __restrict__ bias, int bias_f32,
                                                             const f16* __restrict__ residual, long MN, int N, int splits, int batch, long strideC, int act,
                                                             const f16* __restrict__ rowbias, int rb_rows, long rb_ld, long ldc, f16* __restrict__ C2, long ldc2) {
    osg_pin_all(partial, C, bias, bias_f32, residual, MN, N, splits, batch, strideC, act, rowbias, rb_rows, rb_ld, ldc, C2, ldc2);
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long total = MN * batch;
    if (idx >= total) return;
    const int b = (int)(idx / MN);
    const long e = idx - (long)b * MN;
    const long m = e / N;
    const int n = (int)(e - m * N);
    // round 6: the epilogue operands by UNCONDITIONAL loads (an absent operand reads the slab instead and is replaced by -0.0, the exact neutral element of the
    // additions below): loaded inside `if (bias) ...` blocks, every operand cost a wait at the end of its block -- a dependent trip to memory in front of the slab
    // loads of a launch that lasts ~5 us.  Same additions in the same order => the same bits.
    const bool has_b32 = bias && bias_f32, has_b16 = bias && !bias_f32;
    const f16* safe = reinterpret_cast<const f16*>(partial);
    const f32x4 bv_l = *reinterpret_cast<const f32x4*>(has_b32 ? (const float*)bias + n : partial);
    const f16x4 b16_l = *reinterpret_cast<const f16x4*>(has_b16 ? (const f16*)bias + n : safe);
    const f16x4 rb_l = *reinterpret_cast<const f16x4*>(rowbias ? rowbias + (m / (rb_rows > 0 ? rb_rows : 1)) * rb_ld + n : safe);
    const f16x4 rv_l = *reinterpret_cast<const f16x4*>(residual ? residual + b * strideC + e : safe);
    const float* src = partial + (long)b * splits * MN + e;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < splits; s0 += SB) {
        f32x4 part[SB];
#pragma unroll
        for (int u = 0; u < SB; u++) part[u] = *reinterpret_cast<const f32x4*>(src + (long)min(s0 + u, splits - 1) * MN);
#pragma unroll
        for (int u = 0; u < SB; u++)
            if (s0 + u < splits) v += part[u];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        v[r] += bias ? (has_b32 ? bv_l[r] : (float)b16_l[r]) : -0.0f;
        v[r] += rowbias ? (float)rb_l[r] : -0.0f;
        v[r] += residual ? (float)rv_l[r] : -0.0f;
    }
    f16x4 o;
#pragma unroll
    for (int r = 0; r < 4; r++) o[r] = (f16)osg_apply_act(v[r], act);
    if (ldc == N && !C2) *reinterpret_cast<f16x4*>(C + b * strideC + e) = o;
    else {
        *reinterpret_cast<f16x4*>(C + m * ldc + n) = o;
        if (C2) *reinterpret_cast<f16x4*>(C2 + m * ldc2 + n) = o;
    }
}

template <int BM, int BN, int BK, bool CONV, bool VEC>
int launch_cfg(osg_ctx* ctx, const GemmParams& p, int batch) {
    constexpr int LDS = BK + 8;
    size_t smem = (size_t)2 * (BM + BN) * LDS * sizeof(f16);
    auto kern = gemm_kernel<BM, BN, BK, CONV, VEC>;
    static unsigned long long attr_mask = 0;   // (per device: hipFuncSetAttribute is, and a process may hold several)
    if (osg_first_on_device(attr_mask) && smem > 48 * 1024) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, batch * p.splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, ctx->compute, p);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}


// v2 tile / ring / split-K choice.  Measured on MI355X (tools/gemm_probe.py): the L2->LDS DMA path sustains ~23 B/clk per CU
// and bounds every configuration (a 128x128x64 k-tile moves 32 KiB for 515 MFMA cycles), so the model is: k-tile time =
// max(MFMA, bytes / 23) (+ ~450 exposed cycles when a block is alone on its CU), whole rounds of tiles over the CU slots,
// a fixed fill + epilogue per round, and the extra pass of a split-K reduce.
struct V2Choice { int cfg, nst, splits, ks = 1, fold = 0, spec = 0; };   // ks = 2: two wave groups on alternating k-tiles (gemm2_kernel KS); fold: split-K finished by splitk_fold_acc (no reduce launch); spec: 4 loader waves beside the 4 math waves (gemm2_kernel SPEC; round 6: the 128x128 and 128x160 tiles with a 4-stage ring)
static const int kV2BM[8] = {128, 128, 64, 64, 128, 128, 64, 64}, kV2BN[8] = {128, 64, 64, 128, 160, 80, 80, 160};   // (64x128 and the round-6 tiles 4 .. 7 of osg_gemm_wide.hip: measured candidates only)
// every legal (tile, stages, splits) with its modelled cost in cycles, cheapest first
struct V2Form { bool conv = false, ln1 = false, ln2 = false, geglu = false, rowstats = false, w8 = false; };   // what the launch needs of an instantiation (the round-6 tiles hold a subset)
static std::vector<std::pair<double, V2Choice>> rank_v2(const osg_ctx* ctx, int M, int N, int K, int batch, bool allow_split, V2Form form = V2Form{}) {
    const double cus = ctx->num_cu;
    const int kt = K / 64;
    std::vector<std::pair<double, V2Choice>> out;
    static const bool no_wide = getenv("OSG_TUNE_NO_WIDE") != nullptr;   // (A/B runs: the candidate set of round 5)
    for (int c = 0; c < (ctx->autotune ? (no_wide ? 4 : 8) : 3); c++)
        for (int nst = 8; nst >= 2; nst -= 2) {
            // 6 / 8 stages (every tile of a short-K GEMM in flight at once): only as a measured candidate, only where the ring fits the LDS
            if (form.w8) {
                if (!osg_mm::w8_tile_has(c, nst, form.conv) || (form.geglu && c >= 5)) continue;   // (the WQ = 1 instantiations, osg_gemm_w8.hip)
                if ((nst == 6 || nst == 8) && !ctx->autotune) continue;
            } else if (c < 4) {
                if (nst == 6 && (!ctx->autotune || c == 0)) continue;
                if (nst == 8 && (!ctx->autotune || c != 2)) continue;
            } else if (!osg_mm::wide_tile_has(c, nst, form.conv, form.ln1, form.ln2, form.geglu, form.rowstats)) continue;
            if (c >= 4 && N % 80 != 0) continue;              // (the 80 / 160-column tiles are for the widths they divide)
            const int bnp = (kV2BN[c] + 31) / 32 * 32;
            const double tiles = (double)((M + kV2BM[c] - 1) / kV2BM[c]) * ((N + kV2BN[c] - 1) / kV2BN[c]) * batch;
            const double mfma = kV2BM[c] * kV2BN[c] * 128.0 / 4069.0;
            const double tload = (kV2BM[c] * 128.0 + kV2BN[c] * (form.w8 ? 64.0 : 128.0)) / 23.0;
            const int smem = form.w8 ? nst * (kV2BM[c] * 128 + (kV2BN[c] + 63) / 64 * 64 * 64) : nst * (kV2BM[c] + bnp) * 128;
            if (smem > 160 * 1024) continue;
            const int bpc = std::min(4, 163840 / smem);
            for (int s = 1; s <= (allow_split ? 16 : 1); s++) {
                if (s > 1 && (kt / s < (ctx->autotune ? 3 : 8))) break;   // measured choice: let shorter slices compete too
                const int kts = (kt + s - 1) / s;
                if (s > 1 && (kts * (s - 1) >= kt)) continue;   // an empty split
                const double blocks = tiles * s;
                const double rounds = std::ceil(blocks / (cus * bpc));
                const double conc = std::min((double)bpc, std::ceil(blocks / cus));
                const double tk = conc <= 1.0 ? std::max(mfma, tload) + 450.0 : conc * std::max(mfma, tload);
                double cost = rounds * (kts * tk + 3500.0);
                if (s > 1) cost += 9000.0 + (double)M * N * batch * s * 4.0 / 2000.0;   // reduce launch (measured ~4-7 us) + slab traffic
                out.push_back({cost, V2Choice{c, nst, s}});
                // (measured candidates only, round 6) the one-workgroup-per-CU tiles with their DMA requests issued by four LOADER waves: a wave that issues both
                // stalls ~70-100 cycles per request (the CU's address path takes 1 KiB per ~17 cycles and the four waves queue on it) with its MFMAs behind them
                static const bool no_spec = getenv("OSG_TUNE_NO_SPEC") != nullptr;   // (A/B runs)
                if (ctx->autotune && !no_spec && !no_wide && !form.w8 && (c == 0 || c == 4) && nst == 4 && s == 1 && !form.conv && !form.ln1) out.push_back({cost * 0.9995, V2Choice{c, nst, s, 1, 0, 1}});
                // KS = 2 (measured candidates only): the 64x64 tile with a 2- or 4-stage ring, the 128x64 tile with 2 stages (what the 160 KiB hold), >= 2 k-tiles per slice
                static const bool no_ks2 = getenv("OSG_TUNE_NO_KS2") != nullptr;   // (A/B runs)
                if (ctx->autotune && !no_ks2 && !form.w8 && kts >= 2 && ((c == 2 && (nst == 2 || nst == 4)) || (c == 1 && nst == 2))) out.push_back({cost * 0.999, V2Choice{c, nst, s, 2}});
                // (measured candidates only) 2 .. 4 slices folded by the last arriver of each tile instead of a reduce launch: the tiles of at most 10 accumulator quads per lane
                if (ctx->autotune && s >= 2 && s <= 4 && c != 0 && c != 4 && c != 7 && osg_mm::splitk_fold_mode() != 0) out.push_back({cost * 1.0005, V2Choice{c, nst, s, 1, 1}});
            }
        }
    std::stable_sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    return out;
}
static V2Choice choose_v2(const osg_ctx* ctx, int M, int N, int K, int batch) {
    auto r = rank_v2(ctx, M, N, K, batch, true);
    return r.empty() ? V2Choice{0, 4, 1} : r[0].second;
}

static osg_tune::Key tune_key(const osg_ctx* ctx, int kind, const GemmParams& p, int batch) {
    osg_tune::Key k{};
    k.kind = kind; k.device = 0;   /* (one table for every MI355X of a node: ranks seeded from one file make identical choices) */ k.M = p.M; k.N = p.N; k.K = p.K; k.batch = batch;
    if (kind != 0) { k.H = p.H; k.W = p.W; k.Cin = p.Cin; k.KW = p.KW; k.sh = p.sh; k.sw = p.sw; }
    else k.H = p.lda;
    k.flags = (int)p.act | (p.residual ? 16 : 0) | (p.rowbias ? 32 : 0) | (p.bias_f32 ? 64 : 0) | (p.ln_c1 ? 128 : 0) | (p.rs_in ? 256 : 0) | (p.rs_out ? 512 : 0) | (p.w8 ? 1024 : 0);
    return k;
}
// a launch may be repeated for timing only when it does not consume its own output
static bool tune_safe(const GemmParams& p) { return (const void*)p.C != (const void*)p.A && (const void*)p.C != (const void*)p.residual; }

// launch one configuration (reduce kernel included)
template <bool CONV>
int launch_v2_choice(osg_ctx* ctx, GemmParams p, int batch, V2Choice ch) {
    const int ktiles = p.K / 64;
    int kt_per = (ktiles + ch.splits - 1) / ch.splits;
    p.splits = (ktiles + kt_per - 1) / kt_per;
    p.k_per_split = kt_per * 64;
    p.tickets = nullptr;
    p.fold_acc = 0;
    if (p.splits > 1) {
        size_t need = (size_t)batch * p.splits * p.M * p.N * sizeof(float);
        if (ch.fold && ch.ks == 1 && ch.cfg != 0 && ch.cfg != 4 && ch.cfg != 7) {
            // finished inside the kernel by the last k-slice workgroup of each tile (osg_gemm_common.h splitk_fold_acc); the launch falls back to the reduce
            // launch where the fold does not apply
            const long n_tiles = (long)batch * ((p.M + kV2BM[ch.cfg] - 1) / kV2BM[ch.cfg]) * ((p.N + kV2BN[ch.cfg] - 1) / kV2BN[ch.cfg]);
            need = std::max(need, osg_mm::splitk_fold_route(ctx, p, n_tiles, kV2BM[ch.cfg], kV2BN[ch.cfg]));
        }
        if (osg_ensure_workspace(ctx, need)) return 1;
        p.partial = (float*)ctx->ws;
    }
    // split the operand with more unique bytes across the XCDs (each private L2 then streams its slice from HBM once)
    const double a_unique = CONV ? (double)p.a_bytes : (double)p.M * p.K * 2.0;
    p.n_major = (double)p.N * p.K * (p.w8 ? 1.0 : 2.0) > a_unique;
    int rc;
    if (p.w8) {   // uint8 weight codes: the WQ = 1 instantiations (osg_gemm_w8.hip); a (tile, ring) they do not hold falls back to the 64x64 / 128x128 tile with 4 stages
        if (ch.fold && (ch.cfg == 0 || ch.cfg == 4 || ch.cfg == 7)) p.fold_acc = 0;
        rc = osg_mm::launch_v2_w8(ctx, p, batch, ch.cfg, ch.nst, CONV);
        if (rc == -2) {
            const int c2 = kV2BM[ch.cfg] == 128 ? 0 : 2;
            if (c2 == 0) p.fold_acc = 0;
            rc = osg_mm::launch_v2_w8(ctx, p, batch, c2, 4, CONV);
        }
        if (rc == -2) OSG_FAIL(ctx, "osg_gemm_w8: no kernel takes this form with uint8 weight codes");
        if (rc) return rc;
        if (p.splits > 1 && !p.fold_acc) return launch_splitk_reduce(ctx, p, batch);
        return 0;
    }
    if (ch.cfg >= 4) {   // the round-6 tiles (osg_gemm_wide.hip); a form they do not hold falls back to the 64x64 / 128x128 tile of the same ring
        rc = osg_mm::launch_v2_wide(ctx, p, batch, ch.cfg, ch.nst, CONV, ch.spec);
        if (rc != -2) {
            if (rc) return rc;
            if (p.splits > 1 && !p.fold_acc) return launch_splitk_reduce(ctx, p, batch);
            return 0;
        }
        ch.cfg = kV2BM[ch.cfg] == 128 ? 0 : 2; ch.nst = ch.nst == 2 ? 2 : 4; ch.ks = 1; ch.spec = 0;
        if (p.fold_acc && ch.cfg == 0) p.fold_acc = 0;
    }
    if (p.ln_c1 && ch.cfg == 3) ch.cfg = 2;   // (the folded-LayerNorm variants exist for the first three tiles only)
    if (ch.ks == 2 && (ch.cfg == 1 || ch.cfg == 2) && !(p.ln_c1 && !p.rs_in)) {
        // two wave groups on alternating k-tiles: 64x64 (2- or 4-stage ring) and 128x64 (2 stages)
        const bool t64 = ch.cfg == 2, deep = t64 && ch.nst >= 4;
        if constexpr (!CONV) {
            if (p.ln_c1) {
                const int nch = p.rs_np >> 1;
#define OSG_KS_LN2(NCH_)                                                                                                                        \
    rc = !t64 ? launch_v2<128, 64, 2, false, 0, 0, 2, NCH_, 2>(ctx, p, batch)                                                                   \
              : deep ? launch_v2<64, 64, 4, false, 0, 0, 2, NCH_, 2>(ctx, p, batch) : launch_v2<64, 64, 2, false, 0, 0, 2, NCH_, 2>(ctx, p, batch)
                if (nch <= 5) OSG_KS_LN2(5);
                else if (nch <= 10) OSG_KS_LN2(10);
                else OSG_KS_LN2(20);
#undef OSG_KS_LN2
                return rc;
            }
        }
        rc = !t64 ? launch_v2<128, 64, 2, CONV, 0, 0, 0, 5, 2>(ctx, p, batch)
                  : deep ? launch_v2<64, 64, 4, CONV, 0, 0, 0, 5, 2>(ctx, p, batch) : launch_v2<64, 64, 2, CONV, 0, 0, 0, 5, 2>(ctx, p, batch);
        if (rc) return rc;
        if (p.splits > 1) return launch_splitk_reduce(ctx, p, batch);
        return 0;
    }
    if constexpr (!CONV) {
        if (ch.spec && ch.cfg == 0 && ch.nst == 4 && !(p.ln_c1 && !p.rs_in)) {   // 128x128, four loader waves (round 6)
            p.fold_acc = 0;
            if (p.ln_c1) {
                const int nch = p.rs_np >> 1;
                if (nch <= 5) rc = launch_v2<128, 128, 4, false, 0, 1, 2, 5>(ctx, p, batch);
                else if (nch <= 10) rc = launch_v2<128, 128, 4, false, 0, 1, 2, 10>(ctx, p, batch);
                else rc = launch_v2<128, 128, 4, false, 0, 1, 2, 20>(ctx, p, batch);
            } else rc = launch_v2<128, 128, 4, false, 0, 1>(ctx, p, batch);
            if (rc) return rc;
            if (p.splits > 1) return launch_splitk_reduce(ctx, p, batch);
            return 0;
        }
        if (p.ln_c1 && p.rs_in) {   // LayerNorm folded into the GEMM, row statistics handed over by the producer of A
            const int nch = p.rs_np >> 1;
#define OSG_LN2(NCH_)                                                                                                                                \
    do {                                                                                                                                             \
        if (ch.cfg == 0) rc = ch.nst == 4 ? launch_v2<128, 128, 4, false, 0, 0, 2, NCH_>(ctx, p, batch) : launch_v2<128, 128, 2, false, 0, 0, 2, NCH_>(ctx, p, batch); \
        else if (ch.cfg == 1) rc = ch.nst == 4 ? launch_v2<128, 64, 4, false, 0, 0, 2, NCH_>(ctx, p, batch) : launch_v2<128, 64, 2, false, 0, 0, 2, NCH_>(ctx, p, batch); \
        else rc = ch.nst == 4 ? launch_v2<64, 64, 4, false, 0, 0, 2, NCH_>(ctx, p, batch) : launch_v2<64, 64, 2, false, 0, 0, 2, NCH_>(ctx, p, batch);   \
    } while (0)
            if (nch <= 5) OSG_LN2(5);
            else if (nch <= 10) OSG_LN2(10);
            else OSG_LN2(20);
#undef OSG_LN2
            return rc;
        }
        if (p.ln_c1) {   // ... row statistics accumulated beside the MFMAs
            if (ch.cfg == 0) rc = ch.nst == 4 ? launch_v2<128, 128, 4, false, 0, 0, 1>(ctx, p, batch) : launch_v2<128, 128, 2, false, 0, 0, 1>(ctx, p, batch);
            else if (ch.cfg == 1) rc = ch.nst == 4 ? launch_v2<128, 64, 4, false, 0, 0, 1>(ctx, p, batch) : launch_v2<128, 64, 2, false, 0, 0, 1>(ctx, p, batch);
            else rc = ch.nst == 4 ? launch_v2<64, 64, 4, false, 0, 0, 1>(ctx, p, batch) : launch_v2<64, 64, 2, false, 0, 0, 1>(ctx, p, batch);
            return rc;
        }
    }
    if (ch.cfg == 0) rc = ch.nst == 4 ? launch_v2<128, 128, 4, CONV>(ctx, p, batch) : launch_v2<128, 128, 2, CONV>(ctx, p, batch);
    else if (ch.cfg == 1) rc = ch.nst == 6 ? launch_v2<128, 64, 6, CONV>(ctx, p, batch) : ch.nst == 4 ? launch_v2<128, 64, 4, CONV>(ctx, p, batch) : launch_v2<128, 64, 2, CONV>(ctx, p, batch);
    else if (ch.cfg == 3) rc = ch.nst == 6 ? launch_v2<64, 128, 6, CONV>(ctx, p, batch) : ch.nst == 4 ? launch_v2<64, 128, 4, CONV>(ctx, p, batch) : launch_v2<64, 128, 2, CONV>(ctx, p, batch);
    else rc = ch.nst == 8 ? launch_v2<64, 64, 8, CONV>(ctx, p, batch) : ch.nst == 6 ? launch_v2<64, 64, 6, CONV>(ctx, p, batch) : ch.nst == 4 ? launch_v2<64, 64, 4, CONV>(ctx, p, batch) : launch_v2<64, 64, 2, CONV>(ctx, p, batch);
    if (rc) return rc;
    if (p.splits > 1 && !p.fold_acc) return launch_splitk_reduce(ctx, p, batch);
    return 0;
}

template <bool CONV>
int run_gemm_v2(osg_ctx* ctx, GemmParams p, int batch, const V2Choice* forced) {
    const bool allow_split = p.act != OSG_ACT_GEGLU && !p.ln_c1 && !p.rs_out;   // GEGLU pairing / folded LayerNorm / row statistics live in the tile epilogue
    V2Choice ch;
    const bool env_forced = getenv("OSG_GEMM_CFG") || getenv("OSG_GEMM_SPLITS") || getenv("OSG_GEMM_NST") || getenv("OSG_GEMM_KS") || getenv("OSG_GEMM_FOLD") || getenv("OSG_GEMM_SPEC");
    if (forced) {
        ch = *forced;
    } else if (ctx->autotune && !env_forced) {
        const osg_tune::Key key = tune_key(ctx, CONV ? 2 : 0, p, batch);
        osg_tune::Choice tc;
        if (osg_tune::lookup(key, &tc)) {
            ch = {tc.cfg & 7, tc.nst, tc.splits, (tc.cfg & 8) ? 2 : 1, (tc.cfg & 16) ? 1 : 0, (tc.cfg & 32) ? 1 : 0};
        } else {
            V2Form form;
            form.conv = CONV; form.ln1 = p.ln_c1 && !p.rs_in; form.ln2 = p.ln_c1 && p.rs_in; form.geglu = p.act == OSG_ACT_GEGLU; form.rowstats = p.rs_out != nullptr; form.w8 = p.w8 != 0;
            auto ranked = rank_v2(ctx, p.M, p.N, p.K, batch, allow_split, form);
            ch = ranked.empty() ? V2Choice{0, 4, 1} : ranked[0].second;
            if (!ctx->capturing && tune_safe(p) && !ranked.empty() && !osg_tune::frozen()) {
                float best = -1.f;
                for (auto& cand : ranked) {
                    const float us = osg_tune::time_us(ctx, [&] { return launch_v2_choice<CONV>(ctx, p, batch, cand.second); });
                    static const bool dump = getenv("OSG_TUNE_DUMP") != nullptr;
                    if (dump) fprintf(stderr, "[tune] %s M=%d N=%d K=%d flags=%d: tile %dx%d nst=%d splits=%d ks=%d -> %.2f us (model %.0f)\n", CONV ? "conv" : "gemm", p.M, p.N, p.K, key.flags,
                                      kV2BM[cand.second.cfg], kV2BN[cand.second.cfg], cand.second.nst, cand.second.splits, cand.second.ks + 10 * cand.second.fold + 100 * cand.second.spec, us, cand.first);
                    if (us >= 0.f && (best < 0.f || us < best)) { best = us; ch = cand.second; }
                }
                if (best < 0.f) OSG_FAIL(ctx, "osg_gemm: autotune could not time any configuration");
                osg_tune::store(key, osg_tune::Choice{0, ch.cfg | (ch.ks == 2 ? 8 : 0) | (ch.fold ? 16 : 0) | (ch.spec ? 32 : 0), ch.nst, ch.splits, 0, best});
            } else if (osg_tune::frozen())
                osg_tune::remember(key, osg_tune::Choice{0, ch.cfg | (ch.ks == 2 ? 8 : 0) | (ch.fold ? 16 : 0) | (ch.spec ? 32 : 0), ch.nst, ch.splits, 0, -1.f});
        }
    } else {
        if (p.w8) {
            V2Form form;
            form.conv = CONV; form.geglu = p.act == OSG_ACT_GEGLU; form.w8 = true;
            auto r = rank_v2(ctx, p.M, p.N, p.K, batch, allow_split, form);
            ch = r.empty() ? V2Choice{0, 4, 1} : r[0].second;
        } else
        ch = choose_v2(ctx, p.M, p.N, p.K, batch);
        if (const char* e = getenv("OSG_GEMM_CFG")) ch.cfg = atoi(e);
        if (const char* e = getenv("OSG_GEMM_SPLITS")) ch.splits = atoi(e);
        if (!allow_split) ch.splits = 1;
        if (const char* e = getenv("OSG_GEMM_NST")) ch.nst = atoi(e);
        if (const char* e = getenv("OSG_GEMM_KS")) ch.ks = atoi(e) == 2 ? 2 : 1;
        if (const char* e = getenv("OSG_GEMM_SPEC")) ch.spec = atoi(e) != 0;   // (tests / probes: four loader waves, tiles 0 and 4 with a 4-stage ring)
        if (const char* e = getenv("OSG_GEMM_FOLD")) ch.fold = atoi(e) != 0;   // (tests / probes: finish a forced split inside the kernel)
    }
    return launch_v2_choice<CONV>(ctx, p, batch, ch);
}

template <bool CONV>
int run_gemm(osg_ctx* ctx, GemmParams p, int batch, const V2Choice* forced = nullptr) {
    {
        static const bool force_v1 = getenv("OSG_GEMM_V1") != nullptr;
        const bool shape_ok = p.K % 64 == 0 && (CONV ? p.Cin % 64 == 0 : p.lda % 8 == 0);
        const bool align_ok = (((uintptr_t)p.A | (uintptr_t)p.Bt) & 15) == 0 && (p.strideA % 8 == 0) && (p.strideB % 8 == 0);
        const double a_ext = CONV ? (double)p.a_bytes_l : ((double)(p.M - 1) * p.lda + p.K) * 2.0;
        const double b_ext = (double)p.N * p.K * (p.w8 ? 1.0 : 2.0);
        if (p.w8 && !(shape_ok && align_ok && a_ext < 2147483648.0 && b_ext < 2147483648.0))
            OSG_FAIL(ctx, "osg_gemm_w8 / osg_conv2d_nhwc_w8: K (Cin) must be a multiple of 64, the operands 16-byte aligned and smaller than 2 GiB");
        if ((!force_v1 || p.w8) && shape_ok && align_ok && a_ext < 2147483648.0 && b_ext < 2147483648.0) {
            p.a_bytes = (unsigned)a_ext;
            p.b_bytes = (unsigned)b_ext;
            return run_gemm_v2<CONV>(ctx, p, batch, forced);
        }
        if (p.act == OSG_ACT_GEGLU) OSG_FAIL(ctx, "osg_gemm: GEGLU epilogue needs 16-byte aligned operands (direct-to-LDS kernel only)");
        if (p.ln_c1 || p.rs_out) OSG_FAIL(ctx, "osg_gemm_ln / osg_gemm_rowstats: need K % 64 == 0 and 16-byte aligned operands (direct-to-LDS kernel only)");
    }
    const bool vec = CONV ? (p.Cin % 8 == 0) : (p.K % 8 == 0 && p.lda % 8 == 0);
    // ---- tile / split-K selection -------------------------------------------------------------------------
    auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * batch; };
    const long cu = ctx->num_cu;
    int cfg;  // 0: 128x128x32, 1: 128x64x32, 2: 64x64x32
    if (tiles(128, 128) >= cu && p.N % 128 == 0) cfg = 0;
    else if (tiles(128, 64) >= cu && p.M >= 128) cfg = 1;
    else cfg = 2;
    const int bm = cfg == 2 ? 64 : 128, bn = cfg == 0 ? 128 : 64;
    long t = tiles(bm, bn);
    int splits = 1;
    const int BK = 32;
    if (t < cu && p.K >= 1024) {
        splits = (int)((2 * cu + t - 1) / t);
        int max_splits = p.K / 256;
        if (splits > max_splits) splits = max_splits;
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    int ktiles = (p.K + BK - 1) / BK;
    int kt_per = (ktiles + splits - 1) / splits;
    splits = (ktiles + kt_per - 1) / kt_per;
    p.splits = splits;
    p.k_per_split = kt_per * BK;
    if (splits > 1) {
        size_t need = (size_t)batch * splits * p.M * p.N * sizeof(float);
        if (osg_ensure_workspace(ctx, need)) return 1;
        p.partial = (float*)ctx->ws;
    }
    int rc;
#define OSG_DISPATCH(BM_, BN_)                                                   \
    (vec ? launch_cfg<BM_, BN_, 32, CONV, true>(ctx, p, batch) : launch_cfg<BM_, BN_, 32, CONV, false>(ctx, p, batch))
    if (cfg == 0) rc = OSG_DISPATCH(128, 128);
    else if (cfg == 1) rc = OSG_DISPATCH(128, 64);
    else rc = OSG_DISPATCH(64, 64);
#undef OSG_DISPATCH
    if (rc) return rc;
    if (splits > 1) return launch_splitk_reduce(ctx, p, batch);
    return 0;
}


// [K,N] -> [N,K] tiled transpose through LDS
__global__ __launch_bounds__(256) void transpose_kn_nk_kernel(const f16* __restrict__ src, f16* __restrict__ dst, int K, int N) {
    __shared__ f16 tile[32][33];
    int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? src[(long)k * N + n] : (f16)0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) dst[(long)n * K + k] = tile[tx][r];
    }
}

}  // namespace

// the reduce launch of a split-K contraction whose output feeds StatSinks (the slabs hold no finished values for the tile epilogues to add up): a workgroup
// owns 128 rows x 64 columns -- thread = (4 columns, one of 16 row lanes), 8 rows each -- finishes them like splitk_reduce4_kernel and adds the per-group sums
// of what it stored to the sinks.  The bits of C are those of the flat kernel (same additions in the same order per element).
template <int SB>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ partial, f16* __restrict__ C, const void* __restrict__ bias, int bias_f32,
                                                                  const f16* __restrict__ residual, int M, int N, int splits, int act, const f16* __restrict__ rowbias,
                                                                  int rb_rows, long rb_ld, long ldc, f16* __restrict__ C2, long ldc2, StatSink s0, StatSink s1, int hw,
                                                                  int imgs, int per_xcd) {
    // round 6: a workgroup owns 32 rows x 64 channels (rounds 3-5: 128 rows, eight rows per thread one after the other -- eight dependent trips to memory, 17 us per
    // launch against the 5 us of splitk_reduce4_kernel).  A thread has TWO rows; every load of both -- all slices, the epilogue operands (unconditional: an absent
    // operand reads the slab and enters as -0.0, see splitk_reduce4_kernel) -- is in flight before the first addition.  Per element the slices and operands are added in
    // the old order: the same f16 outputs; the statistics are integer sums of the same rounded values: the same tables.
    osg_pin_all(partial, C, bias, bias_f32, residual, M, N, splits, act, rowbias, rb_rows, rb_ld, ldc, C2, ldc2, hw, imgs, per_xcd);
    constexpr int RPT = 2;
    __shared__ float st[16][64][2];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64, n = n0 + cq * 4;
    const long MN = (long)M * N;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq2[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const bool has_b32 = bias && bias_f32, has_b16 = bias && !bias_f32;
        const f16* safe = reinterpret_cast<const f16*>(partial);
        const f32x4 bv_l = *reinterpret_cast<const f32x4*>(has_b32 ? (const float*)bias + n : partial);
        const f16x4 b16_l = *reinterpret_cast<const f16x4*>(has_b16 ? (const f16*)bias + n : safe);
        f32x4 part[RPT][SB];
        f16x4 rb_l[RPT], rv_l[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int m = min(m0 + rl + 16 * k, M - 1);
            const long e = (long)m * N + n;
            rb_l[k] = *reinterpret_cast<const f16x4*>(rowbias ? rowbias + (long)(m / (rb_rows > 0 ? rb_rows : 1)) * rb_ld + n : safe);
            rv_l[k] = *reinterpret_cast<const f16x4*>(residual ? residual + e : safe);
#pragma unroll
            for (int u = 0; u < SB; u++) part[k][u] = *reinterpret_cast<const f32x4*>(partial + (long)min(u, splits - 1) * MN + e);
        }
#pragma unroll
        for (int k = 0; k < RPT; k++) {
            const int m = m0 + rl + 16 * k;
            if (m >= M) break;
            const long e = (long)m * N + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < SB; u++)
                if (u < splits) v += part[k][u];
            for (int z0 = SB; z0 < splits; z0 += SB) {     // (more than SB slices: the rest the old way)
                f32x4 more[SB];
#pragma unroll
                for (int u = 0; u < SB; u++) more[u] = *reinterpret_cast<const f32x4*>(partial + (long)min(z0 + u, splits - 1) * MN + e);
#pragma unroll
                for (int u = 0; u < SB; u++)
                    if (z0 + u < splits) v += more[u];
            }
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float x = v[r];
                x += bias ? (has_b32 ? bv_l[r] : (float)b16_l[r]) : -0.0f;
                x += rowbias ? (float)rb_l[k][r] : -0.0f;
                x += residual ? (float)rv_l[k][r] : -0.0f;
                o[r] = (f16)osg_apply_act(x, act);
                const float f = (float)o[r];
                cs[r] += f;
                cq2[r] = fmaf(f, f, cq2[r]);
            }
            *reinterpret_cast<f16x4*>(C + (long)m * ldc + n) = o;
            if (C2) *reinterpret_cast<f16x4*>(C2 + (long)m * ldc2 + n) = o;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { st[rl][cq * 4 + r][0] = cs[r]; st[rl][cq * 4 + r][1] = cq2[r]; }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, wn = min(64, N - n0), n_img = m0 / hw;
    const StatSink sk[2] = {s0, s1};
    for (int k = 0; k < 2; k++) {
        if (!sk[k].table) continue;
        const int c_lo = n0 + sk[k].ch_off, c_hi = c_lo + wn;
        const int g = c_lo / sk[k].cpg + lane;
        if (g * sk[k].cpg < c_hi) {
            const int a = max(g * sk[k].cpg, c_lo) - c_lo, b = min((g + 1) * sk[k].cpg, c_hi) - c_lo;
            float S = 0.f, Q = 0.f;
            for (int cc = a; cc < b; cc++)
                for (int pp = 0; pp < 16; pp++) { S += st[pp][cc][0]; Q += st[pp][cc][1]; }
            stat_add(per_xcd, reinterpret_cast<unsigned long long*>(sk[k].table), (long)imgs * sk[k].groups * 2, ((long)n_img * sk[k].groups + g) * 2, S, Q, stat_q_scale((long)hw * sk[k].cpg));
        }
    }
}

int osg_mm::launch_splitk_reduce(osg_ctx* ctx, const GemmParams& p, int batch) {
    long MN = (long)p.M * p.N;
    long total = MN * batch;
    const long ldc_ = p.ldc ? p.ldc : (long)p.N;
    static const bool scalar_only = getenv("OSG_SPLITK_REDUCE_SCALAR") != nullptr;   // (A/B)
    if ((p.sink[0].table || p.sink[1].table) && !ctx->tuning && batch == 1 && p.N % 4 == 0 && (ldc_ & 3) == 0 && (p.ldc2 & 3) == 0 && (p.rb_ld & 3) == 0 &&
        (((uintptr_t)p.C | (uintptr_t)p.C2 | (uintptr_t)p.residual | (uintptr_t)p.rowbias) & 7) == 0 && ((uintptr_t)p.bias & 15) == 0 && p.sink_hw > 0 && p.sink_hw % 32 == 0 &&
        p.M % p.sink_hw == 0) {
        // (the output feeds GroupNorm statistics sinks: the reduce launch is where the finished values are)
        const dim3 grid((unsigned)((p.M + 31) / 32), (unsigned)((p.N + 63) / 64));
        const int imgs = p.M / p.sink_hw, per_xcd = ctx->xcd_ids8 ? 1 : 0;
        if (p.splits <= 4)
            hipLaunchKernelGGL(splitk_reduce_stats_kernel<4>, grid, dim3(256), 0, ctx->compute, p.partial, p.C, p.bias, p.bias_f32, p.residual, p.M, p.N, p.splits, p.act, p.rowbias,
                               p.rb_rows, p.rb_ld, ldc_, p.C2, p.ldc2, p.sink[0], p.sink[1], p.sink_hw, imgs, per_xcd);
        else
            hipLaunchKernelGGL(splitk_reduce_stats_kernel<8>, grid, dim3(256), 0, ctx->compute, p.partial, p.C, p.bias, p.bias_f32, p.residual, p.M, p.N, p.splits, p.act, p.rowbias,
                               p.rb_rows, p.rb_ld, ldc_, p.C2, p.ldc2, p.sink[0], p.sink[1], p.sink_hw, imgs, per_xcd);
        OSG_LAUNCH_CHECK(ctx);
        ctx->sink_fused = true;
        return 0;
    }
    if (!scalar_only && p.N % 4 == 0 && (ldc_ & 3) == 0 && (p.ldc2 & 3) == 0 && (p.strideC & 3) == 0 && (p.rb_ld & 3) == 0 && (((uintptr_t)p.C | (uintptr_t)p.C2 | (uintptr_t)p.residual | (uintptr_t)p.rowbias) & 7) == 0 &&
        ((uintptr_t)p.bias & 15) == 0) {
        const unsigned blocks = (unsigned)((total / 4 + 255) / 256);
        if (p.splits <= 4)
            hipLaunchKernelGGL(splitk_reduce4_kernel<4>, dim3(blocks), dim3(256), 0, ctx->compute, p.partial, p.C, p.bias, p.bias_f32, p.residual, MN, p.N, p.splits, batch, p.strideC,
                               p.act, p.rowbias, p.rb_rows, p.rb_ld, ldc_, p.C2, p.ldc2);
        else
            hipLaunchKernelGGL(splitk_reduce4_kernel<8>, dim3(blocks), dim3(256), 0, ctx->compute, p.partial, p.C, p.bias, p.bias_f32, p.residual, MN, p.N, p.splits, batch, p.strideC,
                               p.act, p.rowbias, p.rb_rows, p.rb_ld, ldc_, p.C2, p.ldc2);
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->compute, p.partial, p.C,
                       p.bias, p.bias_f32, p.residual, MN, p.N, p.splits, batch, p.strideC, p.act, p.rowbias, p.rb_rows, p.rb_ld, p.ldc ? p.ldc : (long)p.N, p.C2, p.ldc2);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}


static int conv2d_route(osg_ctx* ctx, GemmParams& p, int N, int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr);

extern "C" {

int osg_transpose_kn_to_nk(osg_ctx* ctx, osg_dtype dtype, const void* src, void* dst, int K, int N) {
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_transpose_kn_to_nk: only f16 implemented");
    dim3 grid((N + 31) / 32, (K + 31) / 32);
    hipLaunchKernelGGL(transpose_kn_nk_kernel, grid, dim3(256), 0, ctx->compute, (const f16*)src, (f16*)dst, K, N);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_gemm(osg_ctx* ctx, osg_dtype dtype, const void* A, const void* B, int b_is_nk, const void* bias, osg_dtype bias_dtype,
             const void* residual, void* C, int M, int N, int K, int batch, long stride_a, long stride_b, long stride_c,
             osg_act act) {
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_gemm: only f16 arithmetic is implemented on the device");
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) OSG_FAIL(ctx, "osg_gemm: invalid shape of inputs");
    if (bias && bias_dtype != OSG_F16 && bias_dtype != OSG_F32) OSG_FAIL(ctx, "osg_gemm: invalid bias dtype");
    if (act == OSG_ACT_GEGLU && (!b_is_nk || residual || N % 32 || K % 64 || batch != 1))
        OSG_FAIL(ctx, "osg_gemm: the GEGLU epilogue needs a pair-interleaved [N,K] weight, N % 32 == 0, K % 64 == 0, no residual");
    const f16* Bt = (const f16*)B;
    long sb = stride_b;
    if (!b_is_nk) {
        // dynamic [K,N] operand: re-lay it out into scratch, then run the K-contiguous kernel
        int nb = stride_b ? batch : 1;
        size_t bytes = (size_t)nb * K * N * sizeof(f16);
        if (osg_ensure_workspace2(ctx, bytes)) return 1;
        f16* tmp = (f16*)ctx->ws2;
        for (int b = 0; b < nb; b++) {
            dim3 grid((N + 31) / 32, (K + 31) / 32);
            hipLaunchKernelGGL(transpose_kn_nk_kernel, grid, dim3(256), 0, ctx->compute, (const f16*)B + (long)b * stride_b,
                               tmp + (long)b * K * N, K, N);
            OSG_LAUNCH_CHECK(ctx);
        }
        Bt = tmp;
        sb = stride_b ? (long)K * N : 0;
    }
    GemmParams p{};
    p.A = (const f16*)A; p.Bt = Bt; p.C = (f16*)C; p.bias = bias; p.residual = (const f16*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = K;
    p.strideA = stride_a; p.strideB = sb; p.strideC = stride_c;
    p.bias_f32 = bias_dtype == OSG_F32; p.act = act;
    return run_gemm<false>(ctx, p, batch);
}

int osg_gemm_rowstats(osg_ctx* ctx, const void* A, const void* B_nk, const void* bias, osg_dtype bias_dtype, const void* residual, void* C, int M, int N,
                      int K, osg_act act, float* rowstats) {
    if (M <= 0 || N <= 0 || K <= 0) OSG_FAIL(ctx, "osg_gemm_rowstats: invalid shape of inputs");
    if (!rowstats || N % 32 || K % 64 || act == OSG_ACT_GEGLU) OSG_FAIL(ctx, "osg_gemm_rowstats: needs N % 32 == 0, K % 64 == 0, a plain activation");
    if (bias && bias_dtype != OSG_F16 && bias_dtype != OSG_F32) OSG_FAIL(ctx, "osg_gemm_rowstats: invalid bias dtype");
    GemmParams p{};
    p.A = (const f16*)A; p.Bt = (const f16*)B_nk; p.C = (f16*)C; p.bias = bias; p.residual = (const f16*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = K;
    p.bias_f32 = bias_dtype == OSG_F32; p.act = act;
    p.rs_out = rowstats; p.rs_np = N / 32;
    return run_gemm<false>(ctx, p, 1);
}

int osg_gemm_ln(osg_ctx* ctx, const void* x, const void* w_nk_folded, const float* c1, const float* c2, float eps, const float* rowstats,
                const void* residual, void* y, int M, int N, int K, osg_act act) {
    if (M <= 0 || N <= 0 || K <= 0) OSG_FAIL(ctx, "osg_gemm_ln: invalid shape of inputs");
    if (!c1 || !c2) OSG_FAIL(ctx, "osg_gemm_ln: c1 and c2 are required");
    if (K % 64 || N % 4) OSG_FAIL(ctx, "osg_gemm_ln: needs K % 64 == 0, N % 4 == 0");
    if (act == OSG_ACT_GEGLU && (residual || N % 32)) OSG_FAIL(ctx, "osg_gemm_ln: the GEGLU epilogue needs N % 32 == 0 and no residual");
    GemmParams p{};
    p.A = (const f16*)x; p.Bt = (const f16*)w_nk_folded; p.C = (f16*)y; p.bias = c2; p.residual = (const f16*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = K;
    p.bias_f32 = 1; p.act = act;
    p.ln_c1 = c1; p.ln_eps = eps;
    if (rowstats) {
        if (K % 32 || K > 1280) OSG_FAIL(ctx, "osg_gemm_ln: handed-over row statistics need K % 32 == 0, K <= 1280");
        p.rs_in = rowstats; p.rs_np = K / 32;
    }
    return run_gemm<false>(ctx, p, 1);
}

int osg_conv2d_nhwc(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w, const void* bias, osg_dtype bias_dtype,
                    const void* residual, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int sh, int sw,
                    int pt, int pl, int pb, int pr, osg_act act) {
    return osg_conv2d_nhwc_rb(ctx, dtype, x, w, bias, bias_dtype, nullptr, 0, residual, y, N, H, W, Cin, Cout, KH, KW, sh, sw, pt, pl, pb, pr, act);
}

int osg_conv2d_nhwc_rb(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w, const void* bias, osg_dtype bias_dtype,
                       const void* image_bias, long image_bias_ld, const void* residual, void* y, int N, int H, int W, int Cin, int Cout,
                       int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act) {
    return osg_conv2d_nhwc_v(ctx, dtype, x, w, bias, bias_dtype, image_bias, image_bias_ld, residual, y, 0, nullptr, 0, N, H, W, Cin, Cout, KH, KW, sh, sw, pt, pl,
                             pb, pr, act);
}

int osg_set_stat_sinks(osg_ctx* ctx, void* table0, int groups0, int cpg0, int ch_off0, void* table1, int groups1, int cpg1, int ch_off1, int rows_per_image) {
    if ((table0 && (groups0 <= 0 || cpg0 <= 0 || ch_off0 < 0)) || (table1 && (groups1 <= 0 || cpg1 <= 0 || ch_off1 < 0)) || rows_per_image <= 0)
        OSG_FAIL(ctx, "osg_set_stat_sinks: invalid argument");
    ctx->pending_sink[0] = osg_ctx::PendingSink{(long long*)table0, groups0, cpg0, ch_off0};
    ctx->pending_sink[1] = osg_ctx::PendingSink{(long long*)table1, groups1, cpg1, ch_off1};
    ctx->pending_hw = rows_per_image;
    return 0;
}

struct W8Quant { int on; float scale; int zp; const float* sc; const float* zv; };   // uint8 weight codes + (scale, zero point): scalars or [N] vectors

static int conv2d_v(osg_ctx* ctx, const W8Quant& q, const void* x, const void* w, const void* bias, osg_dtype bias_dtype,
                    const void* image_bias, long image_bias_ld, const void* residual, void* y, long y_ld, void* y2, long y2_ld, int N, int H, int W,
                    int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act);

int osg_conv2d_nhwc_v(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w, const void* bias, osg_dtype bias_dtype,
                      const void* image_bias, long image_bias_ld, const void* residual, void* y, long y_ld, void* y2, long y2_ld, int N, int H, int W,
                      int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act) {
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_conv2d_nhwc: only f16 arithmetic is implemented on the device");
    return conv2d_v(ctx, W8Quant{}, x, w, bias, bias_dtype, image_bias, image_bias_ld, residual, y, y_ld, y2, y2_ld, N, H, W, Cin, Cout, KH, KW, sh, sw, pt, pl, pb, pr, act);
}

static int w8_check(osg_ctx* ctx, const char* who, const W8Quant& q, int N, int K, const void* a, const void* b) {
    if (K % 64 || ((uintptr_t)a & 15) || ((uintptr_t)b & 15)) { ctx->err = (std::string(who) + ": K (Cin) must be a multiple of 64 and the operands 16-byte aligned"); return 1; }
    if ((q.sc != nullptr) != (q.zv != nullptr) || (q.sc && ((N & 3) || ((uintptr_t)q.sc & 15)))) { ctx->err = (std::string(who) + ": scale / zero-point vectors come as a pair, N % 4 == 0, 16-byte aligned"); return 1; }
    if (!q.sc && (q.zp < 0 || q.zp > 255)) { ctx->err = (std::string(who) + ": zero point outside 0 .. 255"); return 1; }
    return 0;
}

int osg_gemm_w8_v(osg_ctx* ctx, const void* A, const void* Bq_nk, float w_scale, int w_zero_point, const float* w_scale_vec, const float* w_zero_point_vec,
                  const void* bias, osg_dtype bias_dtype, const void* residual, void* C, int M, int N, int K, osg_act act) {
    if (M <= 0 || N <= 0 || K <= 0) OSG_FAIL(ctx, "osg_gemm_w8: invalid shape of inputs");
    if (bias && bias_dtype != OSG_F16 && bias_dtype != OSG_F32) OSG_FAIL(ctx, "osg_gemm_w8: invalid bias dtype");
    if (act == OSG_ACT_GEGLU && (residual || N % 32)) OSG_FAIL(ctx, "osg_gemm_w8: the GEGLU epilogue needs a pair-interleaved [N,K] weight, N % 32 == 0, no residual");
    const W8Quant q{1, w_scale, w_zero_point, w_scale_vec, w_zero_point_vec};
    if (w8_check(ctx, "osg_gemm_w8", q, N, K, A, Bq_nk)) return 1;
    GemmParams p{};
    p.A = (const f16*)A; p.Bt = (const f16*)Bq_nk; p.C = (f16*)C; p.bias = bias; p.residual = (const f16*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = K;
    p.bias_f32 = bias_dtype == OSG_F32; p.act = act;
    p.w8 = 1; p.w_scale = w_scale; p.w_zp = w_zero_point; p.wq_sc = w_scale_vec; p.wq_zp = w_zero_point_vec;
    return run_gemm<false>(ctx, p, 1);
}

int osg_gemm_w8(osg_ctx* ctx, const void* A, const void* Bq_nk, float w_scale, int w_zero_point, const void* bias, osg_dtype bias_dtype,
                const void* residual, void* C, int M, int N, int K, osg_act act) {
    return osg_gemm_w8_v(ctx, A, Bq_nk, w_scale, w_zero_point, nullptr, nullptr, bias, bias_dtype, residual, C, M, N, K, act);
}

int osg_conv2d_nhwc_w8_v(osg_ctx* ctx, const void* x, const void* wq_ohwi, float w_scale, int w_zero_point, const float* w_scale_vec, const float* w_zero_point_vec,
                         const void* bias, osg_dtype bias_dtype, const void* image_bias, long image_bias_ld, const void* residual, void* y, long y_ld, void* y2,
                         long y2_ld, int N, int H, int W, int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act) {
    const W8Quant q{1, w_scale, w_zero_point, w_scale_vec, w_zero_point_vec};
    if (w8_check(ctx, "osg_conv2d_nhwc_w8", q, Cout, Cin, x, wq_ohwi)) return 1;
    return conv2d_v(ctx, q, x, wq_ohwi, bias, bias_dtype, image_bias, image_bias_ld, residual, y, y_ld, y2, y2_ld, N, H, W, Cin, Cout, KH, KW, sh, sw, pt, pl, pb, pr, act);
}

int osg_conv2d_nhwc_w8(osg_ctx* ctx, const void* x, const void* wq_ohwi, float w_scale, int w_zero_point, const void* bias,
                       osg_dtype bias_dtype, const void* image_bias, long image_bias_ld, const void* residual, void* y, int N, int H, int W,
                       int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act) {
    return osg_conv2d_nhwc_w8_v(ctx, x, wq_ohwi, w_scale, w_zero_point, nullptr, nullptr, bias, bias_dtype, image_bias, image_bias_ld, residual, y, 0, nullptr, 0, N, H, W,
                                Cin, Cout, KH, KW, sh, sw, pt, pl, pb, pr, act);
}

}  // extern "C"

static int conv2d_v(osg_ctx* ctx, const W8Quant& q, const void* x, const void* w, const void* bias, osg_dtype bias_dtype,
                    const void* image_bias, long image_bias_ld, const void* residual, void* y, long y_ld, void* y2, long y2_ld, int N, int H, int W,
                    int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act) {
    if (y_ld < 0 || (y_ld && y_ld < Cout) || (y2 && y2_ld < Cout)) OSG_FAIL(ctx, "osg_conv2d_nhwc_v: an output pitch is smaller than Cout");
    if (Cout % 4 == 0 && ((y_ld & 3) || (y2 && (y2_ld & 3)))) OSG_FAIL(ctx, "osg_conv2d_nhwc_v: output pitches must be multiples of 4 elements");
    if ((y_ld && ((uintptr_t)y & 7)) || (y2 && ((uintptr_t)y2 & 7))) OSG_FAIL(ctx, "osg_conv2d_nhwc_v: output views must be 8-byte aligned");
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0)
        OSG_FAIL(ctx, "osg_conv2d_nhwc: invalid argument");
    int Ho = (H + pt + pb - KH) / sh + 1;
    int Wo = (W + pl + pr - KW) / sw + 1;
    if (Ho <= 0 || Wo <= 0) OSG_FAIL(ctx, "osg_conv2d_nhwc: empty output");
    GemmParams p{};
    p.A = (const f16*)x; p.Bt = (const f16*)w; p.C = (f16*)y; p.bias = bias; p.residual = (const f16*)residual;
    p.M = N * Ho * Wo; p.N = Cout; p.K = KH * KW * Cin; p.lda = 0;
    p.bias_f32 = bias_dtype == OSG_F32; p.act = act;
    p.a_bytes_l = (long)N * H * W * Cin * 2;
    p.rowbias = (const f16*)image_bias; p.rb_rows = Ho * Wo; p.rb_ld = image_bias_ld;
    p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.KW = KW; p.sh = sh; p.sw = sw; p.pt = pt; p.pl = pl;
    p.ldc = y_ld == Cout ? 0 : y_ld; p.C2 = (f16*)y2; p.ldc2 = y2 ? y2_ld : 0;
    if (q.on) { p.w8 = 1; p.w_scale = q.scale; p.w_zp = q.zp; p.wq_sc = q.sc; p.wq_zp = q.zv; }
    // GroupNorm statistics of this launch's output (osg_set_stat_sinks): served by the epilogue of the kernel that runs (sink_fused), or by a launch of its own
    bool want_sinks = false;
    for (int k = 0; k < 2; k++) {
        const auto& ps = ctx->pending_sink[k];
        if (ps.table && (k == 0 || y2)) {
            p.sink[k] = StatSink{ps.table, ps.groups, ps.cpg, ps.ch_off};
            want_sinks = true;
        }
        ctx->pending_sink[k] = osg_ctx::PendingSink{};
    }
    p.sink_hw = ctx->pending_hw;
    ctx->sink_fused = false;
    if (want_sinks && (p.sink_hw <= 0 || p.sink_hw != Ho * Wo)) OSG_FAIL(ctx, "osg_conv2d_nhwc_v: the statistics sinks were set for another image size");
    const int rc = conv2d_route(ctx, p, N, Cin, Cout, KH, KW, sh, sw, pt, pl, pb, pr);
    if (rc || !want_sinks || ctx->sink_fused) return rc;
    return osg_mm::launch_colstats(ctx, p.C, p.ldc ? p.ldc : (long)Cout, p.M, Cout, p.sink_hw, p.sink);
}

static int conv2d_route(osg_ctx* ctx, GemmParams& p, int N, int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr) {
    (void)N;
    if (!p.w8 && Cin == 4 && KH == 3 && KW == 3 && Cout % 16 == 0 && (size_t)Cout * (144 + 4) <= 160 * 1024 && (p.ldc % 4) == 0 && (p.ldc2 % 4) == 0 &&
        (!p.rowbias || p.rb_ld % 4 == 0)) {
        const size_t smem = (size_t)Cout * (144 + 4);
        static size_t attr_smem = 0;
        if (smem > 64 * 1024 && smem > attr_smem) {
            OSG_HIP(ctx, hipFuncSetAttribute((const void*)conv_cin4_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_smem = smem;
        }
        hipLaunchKernelGGL(conv_cin4_mfma_kernel, dim3((p.M + 31) / 32), dim3(256), smem, ctx->compute, p);
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    // a 1x1 / stride 1 / no-pad convolution IS a plain GEMM over the pixels
    if (KH == 1 && KW == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0 && pb == 0 && pr == 0) {
        p.lda = Cin;
        return run_gemm<false>(ctx, p, 1);
    }
    if (KH == 3) {
        const bool env_forced = getenv("OSG_CONV3X3_BN") || getenv("OSG_CONV3X3_SPLITS") || getenv("OSG_CONV3X3_DBG") || getenv("OSG_GEMM_CFG") ||
                                getenv("OSG_GEMM_SPLITS") || getenv("OSG_GEMM_NST");
        if (!ctx->autotune || env_forced) {
            int rc3 = osg_conv3x3_run(ctx, p);
            if (rc3 >= 0) return rc3;
        } else if (osg_conv3x3_prepare(ctx, p) == 0) {
            // measured choice between the halo-reuse kernel's (BN, splits) and the implicit-GEMM kernel's (tile, stages, splits)
            const osg_tune::Key key = tune_key(ctx, 1, p, 1);
            osg_tune::Choice tc;
            if (!osg_tune::lookup(key, &tc)) {
                auto r3 = osg_conv3x3_rank(ctx, p);
                tc = osg_tune::Choice{1, 0, 0, r3.empty() ? 1 : r3[0].second.second % 1000, r3.empty() ? 128 : r3[0].second.first, -1.f};
                if (!ctx->capturing && tune_safe(p) && !osg_tune::frozen()) {
                    float best = -1.f;
                    static const bool dump3 = getenv("OSG_TUNE_DUMP") != nullptr;
                    for (auto& c : r3)
                        for (int nl : {4, 8}) {   // (nst of a conv3x3 row = loader waves of the halo kernel)
                            static const bool no8 = getenv("OSG_TUNE_NO_NL8") != nullptr;   // (A/B runs)
                            if (nl == 8 && (no8 || p.w8)) continue;   // (uint8 weight codes: the 4-loader kernel only)
                            const int fold3 = c.second.second >= 1000;   // (osg_conv3x3_rank: splits + 1000 = the same split, folded in the kernel)
                            const int s3 = c.second.second % 1000;
                            const float us = osg_tune::time_us(ctx, [&] { return osg_conv3x3_launch(ctx, p, c.second.first, s3, nl, fold3); });
                            if (dump3) fprintf(stderr, "[tune] conv3x3 N*H*W=%d Cin=%d Cout=%d W=%d: halo bn=%d splits=%d loaders=%d -> %.2f us\n", p.M, p.Cin, p.N, p.W, c.second.first, c.second.second, nl, us);
                            if (us >= 0.f && (best < 0.f || us < best)) { best = us; tc = osg_tune::Choice{1, fold3 ? 16 : 0, nl, s3, c.second.first, us}; }
                        }
                    V2Form form3;
                    form3.conv = true; form3.w8 = p.w8 != 0;
                    auto r2 = rank_v2(ctx, p.M, p.N, p.K, 1, true, form3);
                    if (r2.size() > 6) r2.resize(6);
                    for (auto& c : r2) {
                        const V2Choice ch = c.second;
                        const float us = osg_tune::time_us(ctx, [&] { return run_gemm<true>(ctx, p, 1, &ch); });
                        if (us >= 0.f && (best < 0.f || us < best)) { best = us; tc = osg_tune::Choice{0, ch.cfg | (ch.ks == 2 ? 8 : 0) | (ch.fold ? 16 : 0), ch.nst, ch.splits, 0, us}; }
                    }
                    if (best < 0.f) OSG_FAIL(ctx, "osg_conv2d_nhwc: autotune could not time any configuration");
                    osg_tune::store(key, tc);
                } else if (osg_tune::frozen())
                    osg_tune::remember(key, tc);
            }
            if (tc.family == 1) return osg_conv3x3_launch(ctx, p, tc.bn, tc.splits, tc.nst == 8 ? 8 : 4, (tc.cfg & 16) ? 1 : 0);
            const V2Choice ch{tc.cfg & 7, tc.nst, tc.splits, (tc.cfg & 8) ? 2 : 1, (tc.cfg & 16) ? 1 : 0};
            return run_gemm<true>(ctx, p, 1, &ch);
        }
    }
    return run_gemm<true>(ctx, p, 1);
}
