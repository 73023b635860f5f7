// The ROW-LOCAL tail of a transformer block as ONE launch (round 4).
//
// Everything behind the self-attention of a BasicTransformerBlock depends, per token row, on that row alone -- except cross-attention's K / V,
// which are 77 context tokens shared by every row of an image:
//
//     x1 = attn1.to_out(a1) + x0                          (reference: MatMul + Add + Add,   src/onnxstream.cpp:5669-5861, :3906-4000)
//     q  = attn2.to_q(LayerNorm(x1))                      (the 9-op LayerNorm chain :5237-5604, MatMul)
//     a2 = softmax(q k^T * s) v      per head             (AttentionFusedOps :6696-6929 after the head split / merge Reshape / Transpose ops)
//     x2 = attn2.to_out(a2) + x1
//     h  = GEGLU(ff.net.0.proj(LayerNorm(x2)))            (Slice, Slice, Div, Erf :4001-4139, Add, Mul, Mul, Mul)
//     x3 = ff.net.2(h) + x2
//   [ y  = proj_out(x3) + x_in                            (the 1x1 Conv :4494-4707 behind the block, + the spatial residual Add) ]
//
// The round-3 plan ran this as 7 launches (141 us at the 64x64 level for 27.6 GFLOP = 11 us of matrix work; every launch lasts as long as ONE
// workgroup).  Here ONE workgroup owns 64 token rows from a1 to y:
//   * the row block lives in LDS between the contractions (three [64 x C] f16 images in the k-tiled, XOR-swizzled layout gemm2_kernel uses for its A
//     tiles: fragment reads are the same conflict-free ds_read_b128), the [64 x 4C] GEGLU activation is never formed: it is produced 128 columns at a
//     time and contracted against the matching 128-deep slice of ff.net.2 on the spot (accumulators of x3 stay in registers across the chunks);
//   * the four waves split the OUTPUT columns (64 x 80 per wave at C = 320), so every weight element is read by exactly one wave -- weights go
//     global -> registers directly (16-byte loads, fragment layout, one k-tile = 10 loads ahead), no LDS ring, no barrier inside a contraction;
//   * LayerNorm is the standalone kernel's arithmetic on the rows in LDS; cross-attention runs per (head, 16-row group) on v_mfma_f32_16x16x16_f16
//     (head dim 40 = 2.5 k-steps, probabilities go from accumulator layout straight into the B operand) against K / V^T re-packed once per pass
//     (osg_tblock_kv_pack: zero padding to 80 tokens x 48 channels is in the pack, not in the kernel).
// Numerics: f32 accumulation in the order of gemm2_kernel's one-slice form, one RNE rounding per reference op boundary that survives fusion level 2
// (x1, LN, q, a2, x2, LN, h, x3, y), scores / probabilities in f32 as attn_kernel keeps them.
#include "osg_common.h"

#include <cstdlib>
#include <type_traits>

namespace osg_tb {

struct TailParams {
    const f16 *a1, *x0;
    const f16 *wo1, *bo1;
    const f16 *g2, *be2;
    float eps2;
    const f16 *wq2, *bq2;
    const f16 *kp, *vtp;
    float sc_log2e;
    int Tk;
    const f16 *wo2, *bo2;
    const f16 *g3, *be3;
    float eps3;
    const f16 *w1, *b1, *w2, *b2;
    const f16 *wpo, *bpo, *xin;
    f16 *out, *out2;
    long ldo, ldo2;
    int M, rows_per_img, heads;
    f16* dbg[8];
};

typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kTileBytes = 8192;   // one k-tile of a row block: 64 rows x 128 B (64 halves), chunk c of row r at slot c ^ (r & 7)

// byte offset of element (m, n), n % 4 == 0, inside a swizzled [64 x C] LDS image
__device__ __forceinline__ int img_off(int m, int n) { return (n >> 6) * kTileBytes + m * 128 + ((((n & 63) >> 3) ^ (m & 7)) << 4) + (n & 7) * 2; }

// the B fragments of one 64-deep k-tile of a [N][ld] weight (k contiguous), NT tiles of 16 rows `rstep` BYTES apart: b[ks * NT + j].  Buffer loads:
// the descriptor and the tile / chunk offset `so` are wave-uniform (scalar registers), the ONE per-lane 32-bit offset `lo` (row nb + l16, k 8 g) never
// changes -- no 64-bit pointer arithmetic in vector registers (hipcc turns plain pointer loads into a VGPR pointer pair per tile row)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 ldb(__amdgpu_buffer_rsrc_t rs, unsigned lo, int so) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, so, 0));
}
template <int NT>
__device__ __forceinline__ void load_b(f16x8 (&b)[10], __amdgpu_buffer_rsrc_t rs, unsigned lo, int so, int rstep) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < NT; j++) b[ks * NT + j] = ldb(rs, lo, so + j * rstep + ks * 64);
}
// GEGLU projection: tiles 0, 1 = value rows, tiles 2, 3 = the matching gate rows (gate_off bytes further down)
__device__ __forceinline__ void load_b_geglu(f16x8 (&b)[10], __amdgpu_buffer_rsrc_t rs, unsigned lo, int so, int rstep, int gate_off) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 4; j++) b[ks * 4 + j] = ldb(rs, lo, so + (j & 1) * rstep + (j >> 1) * gate_off + ks * 64);
}

// one k-tile of a [64 x 16 NT] wave tile: A fragments from the swizzled LDS image, B fragments from a register slot
template <int NT>
__device__ __forceinline__ void mma_ktile(const char* At, int a_rd, const f16x8 (&b)[10], f32x4 (&acc)[4][NT]) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        f16x8 a[4];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = *reinterpret_cast<const f16x8*>(At + ((a_rd + i * 2048) ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ks * NT + j], a[i], acc[i][j], 0, 0, 0);
    }
}

// a whole contraction over KT k-tiles.  SS = index of its first k-tile in the kernel-wide sequence of k-tiles: tile s is consumed from register slot
// s & 1 while the loads of tile s + 1 fly into the other one; `load(b, t)` requests this contraction's tile t, `next(b)` the FIRST tile of whatever
// contraction follows.
template <int KT, int NT, int SS, typename Load, typename Next>
__device__ __forceinline__ void gemm_stage(const char* A, int a_rd, f16x8 (&bq)[2][10], f32x4 (&acc)[4][NT], Load&& load, Next&& next) {
#pragma unroll
    for (int t = 0; t < KT; t++) {
        if (t + 1 < KT) load(bq[(SS + t + 1) & 1], t + 1);
        else next(bq[(SS + t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every request down to its first use: one exposed L2 round trip per fragment)
        mma_ktile<NT>(A + t * kTileBytes, a_rd, bq[(SS + t) & 1], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[4][NT]) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc + bias (+ residual from the LDS image `res`) -> f16 -> the LDS image `dst`.  res may be dst: a lane reads exactly the bytes it overwrites.
// (uniform decisions -- is there a bias, a residual -- are taken OUTSIDE the tile loops: inside, hipcc clones the tile code per combination)
template <int NT, bool RES>
__device__ __forceinline__ void epi_to_lds(const f32x4 (&acc)[4][NT], const f16* __restrict__ bias, int nb, const char* res, char* dst, int lane) {
    const int l16 = lane & 15, g4 = (lane >> 4) * 4;
    f16x4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) bv[j] = f16x4{0, 0, 0, 0};
    if (bias) {
#pragma unroll
        for (int j = 0; j < NT; j++) bv[j] = *reinterpret_cast<const f16x4*>(bias + nb + j * 16 + g4);
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int n = nb + j * 16 + g4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int off = img_off(i * 16 + l16, n);
            f16x4 rv = f16x4{0, 0, 0, 0};
            if constexpr (RES) rv = *reinterpret_cast<const f16x4*>(res + off);
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (f16)((acc[i][j][r] + (float)bv[j][r]) + (float)rv[r]);
            *reinterpret_cast<f16x4*>(dst + off) = o;
        }
    }
}

// acc + bias + residual (RES 1: LDS image, 2: global rows ldr apart) -> f16 -> global rows (and, OUT2, a second destination)
template <int NT, int RES, bool OUT2>
__device__ __forceinline__ void epi_to_global(const f32x4 (&acc)[4][NT], const f16* __restrict__ bias, int nb, const char* res_lds, const f16* __restrict__ res_g, long ldr,
                                              f16* __restrict__ out, long ldo, f16* __restrict__ out2, long ldo2, long row0, int lane) {
    const int l16 = lane & 15, g4 = (lane >> 4) * 4;
    f16x4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) bv[j] = f16x4{0, 0, 0, 0};
    if (bias) {
#pragma unroll
        for (int j = 0; j < NT; j++) bv[j] = *reinterpret_cast<const f16x4*>(bias + nb + j * 16 + g4);
    }
    f16x4 rv[4][NT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int m = i * 16 + l16, n = nb + j * 16 + g4;
            if constexpr (RES == 1) rv[i][j] = *reinterpret_cast<const f16x4*>(res_lds + img_off(m, n));
            else if constexpr (RES == 2) rv[i][j] = *reinterpret_cast<const f16x4*>(res_g + (row0 + m) * ldr + n);
            else rv[i][j] = f16x4{0, 0, 0, 0};
        }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int m = i * 16 + l16, n = nb + j * 16 + g4;
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (f16)((acc[i][j][r] + (float)bv[j][r]) + (float)rv[i][j][r]);
            *reinterpret_cast<f16x4*>(out + (row0 + m) * ldo + n) = o;
            if constexpr (OUT2) *reinterpret_cast<f16x4*>(out2 + (row0 + m) * ldo2 + n) = o;
        }
}

// LayerNorm of the 64 rows of image X into image P: layer_norm_kernel's arithmetic (osg_norm.hip), four adjacent lanes per row
template <int C>
__device__ __forceinline__ void ln_rows(const char* X, char* P, const f16* __restrict__ gamma, const f16* __restrict__ beta, float eps, int tid) {
    constexpr int NCH = C / 8, PER = NCH / 4;
    static_assert(NCH % 4 == 0, "row chunks split over four lanes");
    const int row = tid >> 2, part = tid & 3;
    float v[PER][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = part + 4 * i;
        const f16x8 t = *reinterpret_cast<const f16x8*>(X + (c >> 3) * kTileBytes + row * 128 + (((c & 7) ^ (row & 7)) << 4));
#pragma unroll
        for (int e = 0; e < 8; e++) { v[i][e] = (float)t[e]; s += v[i][e]; }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) { const float d = v[i][e] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = part + 4 * i;
        const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + c * 8);
        const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + c * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (f16)((v[i][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
        *reinterpret_cast<f16x8*>(P + (c >> 3) * kTileBytes + row * 128 + (((c & 7) ^ (row & 7)) << 4)) = o;
    }
}

template <int C>
__device__ __forceinline__ void dump_img(const char* img, f16* __restrict__ dst, long row0, int tid) {
    if (!dst) return;
    constexpr int NCH = C / 8;
    for (int idx = tid; idx < 64 * NCH; idx += 256) {
        const int row = idx / NCH, c = idx - row * NCH;
        *reinterpret_cast<f16x8*>(dst + (row0 + row) * C + c * 8) =
            *reinterpret_cast<const f16x8*>(img + (c >> 3) * kTileBytes + row * 128 + (((c & 7) ^ (row & 7)) << 4));
    }
}

// cross-attention of this wave's heads over the 64 rows: Q from image `Qi`, output into image `Oi`.  Packs: K [head][TKT*16][DP], V^T [head][DP][TKT*16]
template <int D, int TKT>
__device__ __forceinline__ void cross_attention(const char* Qi, char* Oi, const f16* __restrict__ kp, const f16* __restrict__ vtp, int head0, int nheads, float c, int Tk, int lane) {
    constexpr int DS = (D + 15) / 16, DP = DS * 16, TKP = TKT * 16;
    const int l16 = lane & 15, g = lane >> 4;
    for (int hh = 0; hh < nheads; hh++) {
        const int h = head0 + hh;
        const f16* kh = kp + (long)h * TKP * DP;
        const f16* vh = vtp + (long)h * DP * TKP;
        f16x4 kf[TKT][DS], vf[DS][TKT];
#pragma unroll
        for (int tt = 0; tt < TKT; tt++)
#pragma unroll
            for (int ds = 0; ds < DS; ds++) kf[tt][ds] = *reinterpret_cast<const f16x4*>(kh + (tt * 16 + l16) * DP + ds * 16 + g * 4);
#pragma unroll
        for (int dt = 0; dt < DS; dt++)
#pragma unroll
            for (int tt = 0; tt < TKT; tt++) vf[dt][tt] = *reinterpret_cast<const f16x4*>(vh + (dt * 16 + l16) * TKP + tt * 16 + g * 4);
#pragma unroll 1
        for (int mi = 0; mi < 4; mi++) {
            const int m = mi * 16 + l16;
            f16x4 qf[DS];
#pragma unroll
            for (int ds = 0; ds < DS; ds++) {
                const int dd = ds * 16 + g * 4;
                const bool ok = dd < D;                          // (D % 4 == 0: a fragment is all inside or all outside the head)
                const f16x4 t = *reinterpret_cast<const f16x4*>(Qi + img_off(m, h * D + (ok ? dd : 0)));
                qf[ds] = ok ? t : f16x4{0, 0, 0, 0};
            }
            f32x4 s[TKT];
#pragma unroll
            for (int tt = 0; tt < TKT; tt++) {
                s[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ds = 0; ds < DS; ds++) s[tt] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[tt][ds], qf[ds], s[tt], 0, 0, 0);
            }
            // lane holds tokens tt*16 + 4g + r of row m: scale (log2 domain), mask the padding tokens, row maximum over its 4 lane groups
            float mx = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < TKT; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = (tt * 16 + g * 4 + r) < Tk ? s[tt][r] * c : -INFINITY;
                    s[tt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
            f16x4 pf[TKT];
#pragma unroll
            for (int tt = 0; tt < TKT; tt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = __builtin_amdgcn_exp2f(s[tt][r] - mx);
                    sum += e;
                    pf[tt][r] = (f16)e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int dt = 0; dt < DS; dt++) {
                f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < TKT; tt++) o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[dt][tt], pf[tt], o, 0, 0, 0);
                const int dd = dt * 16 + g * 4;
                if (dd < D) {
                    f16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; r++) ov[r] = (f16)(o[r] * inv);
                    *reinterpret_cast<f16x4*>(Oi + img_off(m, h * D + dd)) = ov;
                }
            }
        }
    }
}

template <int C, int D, int TKT>
__global__ __launch_bounds__(256) void tblock_tail_kernel(TailParams p) {
    constexpr int KT = C / 64;                 // k-tiles of a C-deep contraction
    constexpr int NTW = C / 64;                // 16-column tiles per wave when four waves split C output columns
    constexpr int F = 4 * C;                   // GEGLU hidden width
    constexpr int HC = 128, NCHUNK = F / HC;   // hidden columns per chunk
    constexpr int IMG = KT * kTileBytes;
    static_assert(C % 64 == 0 && NTW <= 5 && NCHUNK % 2 == 0 && 2 * (HC / 64) * kTileBytes <= IMG, "shape");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const X = lds;                       // the residual stream of the row block (x0 -> x1 -> x2)
    char* const P = lds + IMG;                 // the current A operand (a1 -> LN(x1) -> a2 -> LN(x2) -> x3)
    char* const R = lds + 2 * IMG;             // q; then the two GEGLU chunk buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * 64;
    const int img = (int)(row0 / p.rows_per_img);
    const int a_rd = l16 * 128 + ((g ^ (l16 & 7)) << 4);
    const int nb = wave * (C / 4);             // this wave's output columns of a C-wide contraction

    f16x8 bq[2][10];
    // per-lane weight offsets (bytes; the bases stay in scalar registers): row nb + l16 (+ 16 j), k offset 8 g
    const unsigned lo_c = (unsigned)(((nb + l16) * C + g * 8) * 2);          // [C][C] weights
    const unsigned lo_1 = (unsigned)(((wave * 32 + l16) * C + g * 8) * 2);   // w1 [2F][C]: + chunk * HC rows; gate rows F rows further
    const unsigned lo_2 = (unsigned)(((nb + l16) * F + g * 8) * 2);          // w2 [C][F]: + chunk * HC columns
    constexpr int RS_C = 16 * C * 2, RS_F = 16 * F * 2, GATE = F * C * 2, CH1 = HC * C * 2, CH2 = HC * 2;
    const __amdgpu_buffer_rsrc_t u_o1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo1, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_q2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wq2, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_o2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo2, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 2 * F * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, C * F * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_po = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpo, 0, p.wpo ? C * C * 2 : 0, 0x00020000);
    const bool has_po = p.wpo != nullptr;

    load_b<NTW>(bq[0], u_o1, lo_c, 0, RS_C);      // k-tile 0 of the first contraction: in flight while the row block arrives

    // ---- the row block: a1 -> P, x0 -> X (LDS-DMA, the swizzle on the source side as in gemm2_kernel) ----------------------------------
    {
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a1 + row0 * C), 0, 64 * C * 2, 0x00020000);
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x0 + row0 * C), 0, 64 * C * 2, 0x00020000);
        const int rsub = lane >> 3, gch = (lane & 7) ^ rsub;
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
            for (int qq = 0; qq < 2; qq++) {
                const int q8 = qq * 4 + wave;                               // 8-row group inside the tile
                const unsigned off = (unsigned)(((q8 * 8 + rsub) * C + kt * 64 + gch * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(P + kt * kTileBytes + q8 * 1024), 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(X + kt * kTileBytes + q8 * 1024), 16, off, 0, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- x1 = to_out(a1) + x0 ---------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[4][NTW];
        zero_acc(acc);
        gemm_stage<KT, NTW, 0>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { load_b<NTW>(b, u_o1, lo_c, t * 128, RS_C); }, [&](f16x8 (&b)[10]) __attribute__((always_inline)) { load_b<NTW>(b, u_q2, lo_c, 0, RS_C); });
        epi_to_lds<NTW, true>(acc, p.bo1, nb, X, X, lane);
    }
    __builtin_amdgcn_s_barrier();
    dump_img<C>(X, p.dbg[0], row0, tid);
    ln_rows<C>(X, P, p.g2, p.be2, p.eps2, tid);
    __builtin_amdgcn_s_barrier();
    dump_img<C>(P, p.dbg[1], row0, tid);

    // ---- q = to_q(LN(x1)) ---------------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[4][NTW];
        zero_acc(acc);
        gemm_stage<KT, NTW, KT>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { load_b<NTW>(b, u_q2, lo_c, t * 128, RS_C); }, [&](f16x8 (&b)[10]) __attribute__((always_inline)) { load_b<NTW>(b, u_o2, lo_c, 0, RS_C); });
        epi_to_lds<NTW, false>(acc, p.bq2, nb, nullptr, R, lane);
    }
    __builtin_amdgcn_s_barrier();
    dump_img<C>(R, p.dbg[2], row0, tid);

    // ---- a2 = cross-attention(q, K, V): heads split over the waves -------------------------------------------------------------------
    {
        constexpr int DS = (D + 15) / 16, DP = DS * 16, TKP = TKT * 16;
        const int hpw = p.heads >> 2;
        const long per_img = (long)p.heads * TKP * DP;
        cross_attention<D, TKT>(R, P, p.kp + img * per_img, p.vtp + img * per_img, wave * hpw, hpw, p.sc_log2e, p.Tk, lane);
    }
    __builtin_amdgcn_s_barrier();
    dump_img<C>(P, p.dbg[3], row0, tid);

    // ---- x2 = to_out(a2) + x1 -----------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[4][NTW];
        zero_acc(acc);
        gemm_stage<KT, NTW, 2 * KT>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { load_b<NTW>(b, u_o2, lo_c, t * 128, RS_C); }, [&](f16x8 (&b)[10]) __attribute__((always_inline)) { load_b_geglu(b, u_1, lo_1, 0, RS_C, GATE); });
        epi_to_lds<NTW, true>(acc, p.bo2, nb, X, X, lane);
    }
    __builtin_amdgcn_s_barrier();
    dump_img<C>(X, p.dbg[4], row0, tid);
    ln_rows<C>(X, P, p.g3, p.be3, p.eps3, tid);
    __builtin_amdgcn_s_barrier();
    dump_img<C>(P, p.dbg[5], row0, tid);

    // ---- x3 = ff.net.2(GEGLU(ff.net.0.proj(LN(x2)))) + x2, the hidden activation 128 columns at a time ----------------------------------
    // k-tile sequence: G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(9) G2(8) | G2(9); G1 = 5 tiles (NT 4), G2 = 2 tiles (NT 5)
    f32x4 accY[4][NTW];
    zero_acc(accY);
    {
        constexpr int S0 = 3 * KT;                 // sequence index of G1(0)'s first k-tile
        constexpr int G2T = HC / 64;
        f32x4 accG[4][4];
        // GEGLU epilogue of chunk c into hidden buffer c & 1 (value tiles 0, 1; gate tiles 2, 3)
        auto geglu_store = [&](int c) __attribute__((always_inline)) {
            char* H = R + (c & 1) * (G2T * kTileBytes);
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                const int hc = wave * 32 + jt * 16 + g * 4;          // column inside the chunk
                f16x4 bv = f16x4{0, 0, 0, 0}, bg = f16x4{0, 0, 0, 0};
                if (p.b1) {
                    bv = *reinterpret_cast<const f16x4*>(p.b1 + c * HC + hc);
                    bg = *reinterpret_cast<const f16x4*>(p.b1 + F + c * HC + hc);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (f16)((accG[i][jt][r] + (float)bv[r]) * osg_gelu_erf(accG[i][2 + jt][r] + (float)bg[r]));
                    *reinterpret_cast<f16x4*>(H + img_off(i * 16 + l16, hc)) = o;
                }
            }
        };
        // chunk 0: projection only
        zero_acc(accG);
        gemm_stage<KT, 4, S0>(P, a_rd, bq, accG, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { load_b_geglu(b, u_1, lo_1, t * 128, RS_C, GATE); }, [&](f16x8 (&b)[10]) __attribute__((always_inline)) { load_b_geglu(b, u_1, lo_1, CH1, RS_C, GATE); });
        geglu_store(0);
        __builtin_amdgcn_s_barrier();
        // chunk c >= 1: projection of chunk c, then ff.net.2 over chunk c - 1 beside the GEGLU arithmetic of chunk c
        auto ff_iter = [&](auto par, int c, bool last) __attribute__((always_inline)) {
            constexpr int SS = S0 + KT + decltype(par)::value;   // parity of the first k-tile of G1(c): (S0 + KT + 7 (c - 1)) & 1
            const int w1c = c * CH1, w2p = (c - 1) * CH2;   // byte offsets: rows of ff.net.0.proj of chunk c, columns of ff.net.2 of chunk c - 1
            zero_acc(accG);
#pragma unroll
            for (int t = 0; t < KT; t++) {
                if (t + 1 < KT) load_b_geglu(bq[(SS + t + 1) & 1], u_1, lo_1, w1c + (t + 1) * 128, RS_C, GATE);
                else load_b<NTW>(bq[(SS + t + 1) & 1], u_2, lo_2, w2p, RS_F);
                __builtin_amdgcn_sched_barrier(0);
                mma_ktile<4>(P + t * kTileBytes, a_rd, bq[(SS + t) & 1], accG);
                __builtin_amdgcn_sched_barrier(0);
            }
            const char* Hp = R + ((c - 1) & 1) * (G2T * kTileBytes);
#pragma unroll
            for (int u = 0; u < G2T; u++) {
                if (u + 1 < G2T) load_b<NTW>(bq[(SS + KT + u + 1) & 1], u_2, lo_2, w2p + (u + 1) * 128, RS_F);
                else if (!last) load_b_geglu(bq[(SS + KT + u + 1) & 1], u_1, lo_1, w1c + CH1, RS_C, GATE);
                else load_b<NTW>(bq[(SS + KT + u + 1) & 1], u_2, lo_2, w2p + CH2, RS_F);
                __builtin_amdgcn_sched_barrier(0);
                mma_ktile<NTW>(Hp + u * kTileBytes, a_rd, bq[(SS + KT + u) & 1], accY);
            }
            geglu_store(c);
            __builtin_amdgcn_s_barrier();
        };
        static_assert((KT + G2T) % 2 == 1, "chunk parity alternates");
        for (int c = 1; c + 1 < NCHUNK; c += 2) {
            ff_iter(std::integral_constant<int, 0>{}, c, false);
            ff_iter(std::integral_constant<int, 1>{}, c + 1, false);
        }
        ff_iter(std::integral_constant<int, 0>{}, NCHUNK - 1, true);
        // ff.net.2 over the last chunk; its first k-tile was requested by the last ff_iter
        {
            constexpr int SS = S0 + KT + (NCHUNK - 1) * (KT + G2T);          // sequence index of G2(NCHUNK - 1)'s first k-tile
            const char* Hp = R + ((NCHUNK - 1) & 1) * (G2T * kTileBytes);
#pragma unroll
            for (int u = 0; u < G2T; u++) {
                if (u + 1 < G2T) load_b<NTW>(bq[(SS + u + 1) & 1], u_2, lo_2, (NCHUNK - 1) * CH2 + (u + 1) * 128, RS_F);
                else if (has_po) load_b<NTW>(bq[(SS + u + 1) & 1], u_po, lo_c, 0, RS_C);
                __builtin_amdgcn_sched_barrier(0);
                mma_ktile<NTW>(Hp + u * kTileBytes, a_rd, bq[(SS + u) & 1], accY);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (!has_po) {
        if (p.out2) epi_to_global<NTW, 1, true>(accY, p.b2, nb, X, nullptr, 0, p.out, p.ldo, p.out2, p.ldo2, row0, lane);
        else epi_to_global<NTW, 1, false>(accY, p.b2, nb, X, nullptr, 0, p.out, p.ldo, nullptr, 0, row0, lane);
        return;
    }
    // x3 into P (every wave is past its last read of LN(x2): the barrier of the last chunk), then y = proj_out(x3) + x_in
    epi_to_lds<NTW, true>(accY, p.b2, nb, X, P, lane);
    __builtin_amdgcn_s_barrier();
    dump_img<C>(P, p.dbg[6], row0, tid);
    {
        constexpr int SS = 3 * KT + KT + (NCHUNK - 1) * (KT + HC / 64) + HC / 64;
        f32x4 acc[4][NTW];
        zero_acc(acc);
        gemm_stage<KT, NTW, SS>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { load_b<NTW>(b, u_po, lo_c, t * 128, RS_C); }, [&](f16x8 (&)[10]) __attribute__((always_inline)) {});
        if (p.out2) epi_to_global<NTW, 2, true>(acc, p.bpo, nb, nullptr, p.xin, C, p.out, p.ldo, p.out2, p.ldo2, row0, lane);
        else epi_to_global<NTW, 2, false>(acc, p.bpo, nb, nullptr, p.xin, C, p.out, p.ldo, nullptr, 0, row0, lane);
    }
}

// K / V of a cross-attention re-packed for cross_attention(): rows [imgs][Tk] of ONE matrix `base` (row pitch ld), job j's K at column k_col, its V at
// column v_col (head h a further h D columns in) -> kp [img][head][TKP][DP] at dst + job.dst, vtp [img][head][DP][TKP] right behind it; zero padding
struct PackJob { int k_col, v_col, D, dst; };
__global__ __launch_bounds__(256) void kv_pack_kernel(const f16* __restrict__ base, long ld, int Tk, int heads, int imgs, const PackJob* __restrict__ jobs, f16* __restrict__ dst) {
    const int per_job = imgs * heads;
    const int j = blockIdx.x / per_job, ih = blockIdx.x - j * per_job, img = ih / heads, h = ih - img * heads;
    const PackJob job = jobs[j];
    const int D = job.D, DP = (D + 15) / 16 * 16, TKP = 80;
    f16* kp = dst + job.dst + (long)ih * TKP * DP;
    f16* vtp = dst + job.dst + (long)per_job * TKP * DP + (long)ih * TKP * DP;
    const f16* k = base + job.k_col + h * D;
    const f16* v = base + job.v_col + h * D;
    for (int idx = threadIdx.x; idx < TKP * DP; idx += 256) {
        const int t = idx / DP, d = idx - t * DP;
        kp[idx] = (t < Tk && d < D) ? k[((long)img * Tk + t) * ld + d] : (f16)0.f;
    }
    for (int idx = threadIdx.x; idx < TKP * DP; idx += 256) {
        const int d = idx / TKP, t = idx - d * TKP;
        vtp[idx] = (t < Tk && d < D) ? v[((long)img * Tk + t) * ld + d] : (f16)0.f;
    }
}

}   // namespace osg_tb

extern "C" {

int osg_tblock_tail_supported(int M, int rows_per_img, int C, int heads, int Tk) {
    return C == 320 && heads == 8 && M > 0 && M % 64 == 0 && rows_per_img % 64 == 0 && Tk >= 1 && Tk <= 80;
}

size_t osg_tblock_kv_pack_elems(int imgs, int heads, int D) { return (size_t)imgs * heads * 80 * (size_t)((D + 15) / 16 * 16); }

int osg_tblock_kv_pack_jobs(osg_ctx* ctx, const void* base, long ld, int imgs, int Tk, int heads, int njobs, const int* jobs_dev, void* dst) {
    if (Tk < 1 || Tk > 80 || njobs < 1 || imgs < 1 || heads < 1) OSG_FAIL(ctx, "osg_tblock_kv_pack_jobs: unsupported shape");
    hipLaunchKernelGGL(osg_tb::kv_pack_kernel, dim3((unsigned)(njobs * imgs * heads)), dim3(256), 0, ctx->compute, (const f16*)base, ld, Tk, heads, imgs,
                       (const osg_tb::PackJob*)jobs_dev, (f16*)dst);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_tblock_tail(osg_ctx* ctx, const osg_tblock_tail_args* a) {
    if (!osg_tblock_tail_supported(a->M, a->rows_per_img, a->C, a->heads, a->Tk)) OSG_FAIL(ctx, "osg_tblock_tail: unsupported shape (see osg_tblock_tail_supported)");
    if (!(a->scale > 0.f)) OSG_FAIL(ctx, "osg_tblock_tail: scale must be > 0");
    if (!a->a1 || !a->x0 || !a->wo1 || !a->g2 || !a->be2 || !a->wq2 || !a->kp || !a->vtp || !a->wo2 || !a->g3 || !a->be3 || !a->w1 || !a->w2 || !a->out)
        OSG_FAIL(ctx, "osg_tblock_tail: missing operand");
    if (a->wpo && !a->xin) OSG_FAIL(ctx, "osg_tblock_tail: proj_out needs its residual");
    osg_tb::TailParams p;
    p.a1 = (const f16*)a->a1; p.x0 = (const f16*)a->x0;
    p.wo1 = (const f16*)a->wo1; p.bo1 = (const f16*)a->bo1;
    p.g2 = (const f16*)a->g2; p.be2 = (const f16*)a->be2; p.eps2 = a->eps2;
    p.wq2 = (const f16*)a->wq2; p.bq2 = (const f16*)a->bq2;
    p.kp = (const f16*)a->kp; p.vtp = (const f16*)a->vtp;
    p.sc_log2e = a->scale * 1.4426950408889634f;
    p.Tk = a->Tk;
    p.wo2 = (const f16*)a->wo2; p.bo2 = (const f16*)a->bo2;
    p.g3 = (const f16*)a->g3; p.be3 = (const f16*)a->be3; p.eps3 = a->eps3;
    p.w1 = (const f16*)a->w1; p.b1 = (const f16*)a->b1; p.w2 = (const f16*)a->w2; p.b2 = (const f16*)a->b2;
    p.wpo = (const f16*)a->wpo; p.bpo = (const f16*)a->bpo; p.xin = (const f16*)a->xin;
    p.out = (f16*)a->out; p.out2 = (f16*)a->out2;
    p.ldo = a->ldo ? a->ldo : a->C; p.ldo2 = a->ldo2;
    p.M = a->M; p.rows_per_img = a->rows_per_img; p.heads = a->heads;
    for (int i = 0; i < 8; i++) p.dbg[i] = (f16*)a->dbg[i];
    auto kern = osg_tb::tblock_tail_kernel<320, 40, 5>;
    constexpr int smem = 3 * 5 * osg_tb::kTileBytes;
    static bool attr_set = false;
    if (!attr_set) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(a->M / 64)), dim3(256), smem, ctx->compute, p);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}   // extern "C"
