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
// workgroup).  Here ONE workgroup owns a block of token rows from a1 to y -- 64 rows in the text below (RT = 4 row tiles per wave), 32 in the
// variant the library picks while 64-row blocks would leave CUs without one (RT = 2: 50 KB of code instead of 84, one workgroup per CU at M = 8 192):
//   * the row block lives in LDS between the contractions (three [64 x C] f16 images in the k-tiled, XOR-swizzled layout gemm2_kernel uses for its A
//     tiles: fragment reads are the same conflict-free ds_read_b128), the [64 x 4C] GEGLU activation is never formed: it is produced 128 columns at a
//     time and contracted against the matching 128-deep slice of ff.net.2 on the spot (accumulators of x3 stay in registers across the chunks);
//   * the four waves split the OUTPUT columns (64 x 80 per wave at C = 320), so every weight element is read by exactly one wave -- weights go
//     global -> registers directly (16-byte loads, fragment layout, one k-tile = 10 loads ahead), no LDS ring, no barrier inside a contraction;
//   * LayerNorm is the standalone kernel's arithmetic on the rows in LDS; cross-attention runs per (head, 16-row group) on v_mfma_f32_16x16x16_f16
//     (head dim 40 = 2.5 k-steps, probabilities go from accumulator layout straight into the B operand) against K / V^T re-packed once per pass
//     (osg_tblock_kv_pack: zero padding to 80 tokens x 48 channels is in the pack, not in the kernel).
// Weight layout (round 4, second version): [K/8][N][8] ("kn8", osg_tblock_pack_weight) -- for every 8-deep k chunk the N rows sit side by side, 16 bytes each.
// A fragment request (lane l: row l & 15, k chunk l >> 4) is then four runs of 256 contiguous bytes.  Read out of the resident [N][K] layout the same
// request touches 64 different 64-byte pieces in an order in which no four consecutive lanes are contiguous: the CU's address path took ~64 cycles per
// request instead of 16, four waves x 10 requests per k-tile = 2 500 cycles against 640 cycles of MFMA work -- the first version ran as long as the seven
// launches it replaced (profiles/r04_tblock_tail_v1_ab.txt), whatever the prefetch depth.
// Numerics: f32 accumulation in the order of gemm2_kernel's one-slice form, one RNE rounding per reference op boundary that survives fusion level 2
// (x1, LN, q, a2, x2, LN, h, x3, y), scores / probabilities in f32 as attn_kernel keeps them.
// Every workgroup barrier of this file is lds_barrier() (s_waitcnt lgkmcnt(0) + s_barrier): a bare s_barrier does not wait for LDS writes on gfx950.
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
    int pf_sleep0, pf_sleep;   // prefetching workgroups: pauses (x 64 cycles) behind the first contractions' weights and between the feed-forward chunks
    f16* dbg[8];
};

typedef __attribute__((address_space(3))) void* lds_ptr;

// Workgroup barrier behind LDS writes.  A bare s_barrier does not wait for the wave's own outstanding LDS operations (gfx90a and later back off at barriers, so
// hipcc inserts no s_waitcnt in front of one), and LDS requests of different SIMDs are not served in issue order: without the wait a wave past the barrier can
// read what another wave has issued but the LDS has not yet written -- seen as run-to-run differences of round 4's osg_qattn (removed in round 6) at 10 heads of 64 on cold operands
// (profiles/r04_qattn_lds_barrier_race.txt).  lgkmcnt only: vector-memory requests (weight fragments in flight) are deliberately NOT waited for.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}


// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): loop indices that ARE constants (not merely become constants once an unrolling pass
// has run): the register slots bq[s % NS] below must be addressed with literal indices, or the array is left in scratch memory
template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for_impl(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) { static_for_impl<0, N>(f); }

// one k-tile of a row block of 16 RT rows: rows x 128 B (64 halves), chunk c of row r at slot c ^ (r & 7).  RT = 16-row tiles per wave: 4 (64-row blocks) or 2 (32)
template <int RT>
constexpr int kTileBytes = RT * 2048;

// byte offset of element (m, n), n % 4 == 0, inside a swizzled [16 RT x C] LDS image
template <int RT>
__device__ __forceinline__ int img_off(int m, int n) { return (n >> 6) * kTileBytes<RT> + m * 128 + ((((n & 63) >> 3) ^ (m & 7)) << 4) + (n & 7) * 2; }

// the B fragments of one 64-deep k-tile of a kn8 weight ([K/8][N][8]), NT tiles of 16 rows: b[5 ks + j].  Buffer loads: the descriptor and the tile /
// chunk offset `so` are wave-uniform (scalar registers), the ONE per-lane 32-bit offset `lo` ((k chunk g) * N + row nb + l16, times 16 bytes) never changes.
// kstep = bytes between the two 32-deep halves of a tile = 4 N 16.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 ldb(__amdgpu_buffer_rsrc_t rs, unsigned lo, int so) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, so, 0));
}
template <int NT>
__device__ __forceinline__ void load_b(f16x8 (&b)[10], __amdgpu_buffer_rsrc_t rs, unsigned lo, int so, int kstep) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < NT; j++) b[ks * 5 + j] = ldb(rs, lo, so + j * 256 + ks * kstep);
}
// GEGLU projection: tiles 0, 1 = value rows, tiles 2, 3 = the matching gate rows (gate_off bytes further along the row axis).  Same slot layout as five tiles
// (b[5 ks + j]), and the fifth entry is WRITTEN too (a copy of the fourth): where a uniform run-time branch requests either kind of tile into one slot, both
// sides then store the same ten entries in the same order -- otherwise hipcc's store sinking merges the tails of the two sides into a store through a
// pointer phi and the whole slot array stays in scratch memory (480 bytes of private memory per lane, every fragment a scratch round trip)
__device__ __forceinline__ void load_b_geglu(f16x8 (&b)[10], __amdgpu_buffer_rsrc_t rs, unsigned lo, int so, int kstep, int gate_off) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
        for (int j = 0; j < 4; j++) b[ks * 5 + j] = ldb(rs, lo, so + (j & 1) * 256 + (j >> 1) * gate_off + ks * kstep);
        b[ks * 5 + 4] = b[ks * 5 + 3];
    }
}

// one k-tile of a [64 x 16 NT] wave tile: A fragments from the swizzled LDS image, B fragments from a register slot
template <int RT, int NT>
__device__ __forceinline__ void mma_ktile(const char* At, int a_rd, const f16x8 (&b)[10], f32x4 (&acc)[RT][NT]) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        f16x8 a[RT];
#pragma unroll
        for (int i = 0; i < RT; i++) a[i] = *reinterpret_cast<const f16x8*>(At + ((a_rd + i * 2048) ^ (ks << 6)));
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int i = 0; i < RT; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[ks * 5 + j], a[i], acc[i][j], 0, 0, 0);
    }
}

// A whole contraction over KT k-tiles.  Every k-tile of the kernel has an index s in ONE kernel-wide sequence (SS = the index of this contraction's
// first tile): tile s is consumed from register slot s % NS while the requests of the next NS - 1 tiles are in flight -- at tile s the request of tile
// s + NS - 1 goes out, into the slot tile s - 1 has just left.  `load(b, t)` requests this contraction's tile t, `next(b, j)` tile j of whatever follows.
template <int RT, int KT, int NT, int SS, int NS, typename Load, typename Next, typename Mark>
__device__ __forceinline__ void gemm_stage(const char* A, int a_rd, f16x8 (&bq)[NS][10], f32x4 (&acc)[RT][NT], Load&& load, Next&& next, Mark&& mark) {
    static_for<KT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        mark(t);
        if constexpr (t + NS - 1 < KT) load(bq[(SS + t + NS - 1) % NS], t + NS - 1);
        else next(bq[(SS + t + NS - 1) % NS], t + NS - 1 - KT);
        __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks every request down to its first use: one exposed L2 round trip per fragment)
        mma_ktile<RT, NT>(A + t * kTileBytes<RT>, a_rd, bq[(SS + t) % NS], acc);
        __builtin_amdgcn_sched_barrier(0);
    });
}

template <int RT, int KT, int NT, int SS, int NS, typename Load, typename Next>
__device__ __forceinline__ void gemm_stage(const char* A, int a_rd, f16x8 (&bq)[NS][10], f32x4 (&acc)[RT][NT], Load&& load, Next&& next) {
    gemm_stage<RT, KT, NT, SS, NS>(A, a_rd, bq, acc, load, next, [](int) __attribute__((always_inline)) {});
}

// A [C x C] contraction whose KT weight tiles were ALL requested one contraction earlier (slot t = tile t): the weights of the four square contractions of a
// block are cold when the launch starts and only ~200 KB each -- with one or two tiles in flight every one of their k-tiles waits a memory round trip
// (~1 us per tile measured, against 0.4 us for the feed-forward's tiles the prefetching workgroups have long pulled into the L2).  `after(b, t)` runs
// behind tile t's MFMAs: it requests tile t of the NEXT square contraction into the slot that has just been read.
template <int RT, int KT, int NT, typename After, typename Mark>
__device__ __forceinline__ void square_stage(const char* A, int a_rd, f16x8 (&bs)[KT][10], f32x4 (&acc)[RT][NT], After&& after, Mark&& mark) {
    static_for<KT>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        mark(t);
        __builtin_amdgcn_sched_barrier(0);
        mma_ktile<RT, NT>(A + t * kTileBytes<RT>, a_rd, bs[t], acc);
        __builtin_amdgcn_sched_barrier(0);
        after(bs[t], t);
    });
}

template <int RT, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[RT][NT]) {
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// acc + bias (+ residual from the LDS image `res`) -> f16 -> the LDS image `dst`.  res may be dst: a lane reads exactly the bytes it overwrites.
// bias: this op's [C] vector in the LDS copy of the block's small operands (zeros when the op has none) -- a load from global memory here would be a cold
// round trip on the critical path of every stage, and (vector-memory results return in order) a wait for every weight tile requested before it
template <int RT, int NT, bool RES>
__device__ __forceinline__ void epi_to_lds(const f32x4 (&acc)[RT][NT], const char* bias, int nb, const char* res, char* dst, int lane) {
    const int l16 = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int n = nb + j * 16 + g4;
        const f16x4 bv = *reinterpret_cast<const f16x4*>(bias + n * 2);
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const int off = img_off<RT>(i * 16 + l16, n);
            f16x4 rv = f16x4{0, 0, 0, 0};
            if constexpr (RES) rv = *reinterpret_cast<const f16x4*>(res + off);
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (f16)((acc[i][j][r] + (float)bv[r]) + (float)rv[r]);
            *reinterpret_cast<f16x4*>(dst + off) = o;
        }
    }
}

// acc + bias + residual (an LDS image, or fragments `rg` fetched from global rows by the caller) -> f16 -> global rows (and, OUT2, a second destination)
template <int RT, int NT, bool RES_LDS, bool OUT2>
__device__ __forceinline__ void epi_to_global(const f32x4 (&acc)[RT][NT], const char* bias, int nb, const char* res_lds, const f16x4 (&rg)[RT][NT],
                                              f16* __restrict__ out, long ldo, f16* __restrict__ out2, long ldo2, long row0, int lane) {
    const int l16 = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int n = nb + j * 16 + g4;
        const f16x4 bv = *reinterpret_cast<const f16x4*>(bias + n * 2);
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const int m = i * 16 + l16;
            f16x4 rv = rg[i][j];
            if constexpr (RES_LDS) rv = *reinterpret_cast<const f16x4*>(res_lds + img_off<RT>(m, n));
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (f16)((acc[i][j][r] + (float)bv[r]) + (float)rv[r]);
            *reinterpret_cast<f16x4*>(out + (row0 + m) * ldo + n) = o;
            if constexpr (OUT2) *reinterpret_cast<f16x4*>(out2 + (row0 + m) * ldo2 + n) = o;
        }
    }
}

// LayerNorm of the 16 RT rows of image X into image P: layer_norm_kernel's arithmetic (osg_norm.hip), LPR = 256 / rows adjacent lanes per row
template <int RT, int C>
__device__ __forceinline__ void ln_rows(const char* X, char* P, const char* gamma, const char* beta, float eps, int tid) {   // (gamma, beta: LDS copies)
    constexpr int NCH = C / 8, LPR = 16 / RT, PER = NCH / LPR;
    static_assert(NCH % LPR == 0 && (LPR == 4 || LPR == 8), "row chunks split over the lanes of a row");
    const int row = tid / LPR, part = tid % LPR;
    float v[PER][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = part + LPR * i;
        const f16x8 t = *reinterpret_cast<const f16x8*>(X + (c >> 3) * kTileBytes<RT> + row * 128 + (((c & 7) ^ (row & 7)) << 4));
#pragma unroll
        for (int e = 0; e < 8; e++) { v[i][e] = (float)t[e]; s += v[i][e]; }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if constexpr (LPR == 8) s += __shfl_xor(s, 4, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; i++)
#pragma unroll
        for (int e = 0; e < 8; e++) { const float d = v[i][e] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    if constexpr (LPR == 8) q += __shfl_xor(q, 4, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int c = part + LPR * i;
        const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + c * 16);
        const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + c * 16);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (f16)((v[i][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
        *reinterpret_cast<f16x8*>(P + (c >> 3) * kTileBytes<RT> + row * 128 + (((c & 7) ^ (row & 7)) << 4)) = o;
    }
}

template <int RT, int C>
__device__ __forceinline__ void dump_img(const char* img, f16* __restrict__ dst, long row0, int tid) {
    if (!dst) return;
    constexpr int NCH = C / 8;
    for (int idx = tid; idx < 16 * RT * NCH; idx += 256) {
        const int row = idx / NCH, c = idx - row * NCH;
        *reinterpret_cast<f16x8*>(dst + (row0 + row) * C + c * 8) =
            *reinterpret_cast<const f16x8*>(img + (c >> 3) * kTileBytes<RT> + row * 128 + (((c & 7) ^ (row & 7)) << 4));
    }
}

// cross-attention of this wave's heads over the 64 rows: Q from image `Qi`, output into image `Oi`.  Packs: K [head][TKT*16][DP], V^T [head][DP][TKT*16]
template <int RT, int D, int TKT>
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
        for (int mi = 0; mi < RT; mi++) {
            const int m = mi * 16 + l16;
            f16x4 qf[DS];
#pragma unroll
            for (int ds = 0; ds < DS; ds++) {
                const int dd = ds * 16 + g * 4;
                const bool ok = dd < D;                          // (D % 4 == 0: a fragment is all inside or all outside the head)
                const f16x4 t = *reinterpret_cast<const f16x4*>(Qi + img_off<RT>(m, h * D + (ok ? dd : 0)));
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
                    *reinterpret_cast<f16x4*>(Oi + img_off<RT>(m, h * D + dd)) = ov;
                }
            }
        }
    }
}

// one dword of every 128-byte line of [rows x row_bytes] (rows `pitch` bytes apart): pulls the lines into this XCD's L2.  Requests only, nothing waits
// for the data: LDS-DMA loads into a scratch corner of the LDS (no register is written, so no register has to stay reserved until the data arrives)
__device__ __forceinline__ void touch_lines(const void* base, int rows, int row_bytes, long pitch, int tid, char* lds_dummy) {
    const int lpr = (row_bytes + 127) >> 7, n = rows * lpr;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((rows - 1) * pitch + row_bytes), 0x00020000);
    for (int i = tid; i < n; i += 256) {
        const int r = i / lpr, l = i - r * lpr;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)lds_dummy, 4, (unsigned)(r * pitch + l * 128), 0, 0, 0);
    }
}

// NS = register slots for weight tiles (2: one tile ahead, 3: two tiles ahead)
template <int RT, int C, int D, int TKT, int NS>
__global__ __launch_bounds__(256, RT == 2 && NS == 2 ? 2 : 1) void tblock_tail_kernel(TailParams p) {   // (32-row blocks: two workgroups fit a CU -- 72 KB of LDS, <= 256 registers)
    constexpr int RB = 16 * RT;                // rows of a row block
    constexpr int TB = kTileBytes<RT>;
    constexpr int KT = C / 64;                 // k-tiles of a C-deep contraction
    constexpr int NTW = C / 64;                // 16-column tiles per wave when four waves split C output columns
    constexpr int F = 4 * C;                   // GEGLU hidden width
    constexpr int HC = 128, NCHUNK = F / HC;   // hidden columns per chunk
    constexpr int G2T = HC / 64;               // k-tiles of ff.net.2 per chunk
    constexpr int PER = KT + G2T;              // k-tiles per chunk iteration
    constexpr int IMG = KT * TB;
    constexpr int AH = NS - 1;                 // tiles requested ahead
    static_assert(C % 64 == 0 && NTW <= 5 && 2 * G2T * TB <= IMG && (RT == 2 || RT == 4) && (NS == 2 || NS == 3) && PER % NS == 1 && AH <= G2T && AH <= KT, "shape");
    // kn8 byte strides: KS_* = between the two halves of a k-tile (4 k chunks), KT_* = between k-tiles (8 k chunks); N = C for the square weights and ff.net.2
    // ([F/8][C][8]), N = 2F for ff.net.0.proj ([C/8][2F][8])
    constexpr int KS_C = 4 * C * 16, KT_C = 8 * C * 16, KS_1 = 4 * 2 * F * 16, KT_1 = 8 * 2 * F * 16;
    constexpr int GATE = F * 16;               // gate rows sit F rows behind the value rows
    constexpr int CH1 = HC * 16;               // ff.net.0.proj: the rows of the next chunk
    constexpr int CH2 = (HC / 8) * C * 16;     // ff.net.2: the k chunks of the next chunk of hidden columns
    const int tid = threadIdx.x;
    const int nblk = p.M / RB;

    // ---- the workgroups behind the row blocks do no arithmetic: each pulls the block's weights into the L2 of ITS XCD (workgroup i runs on XCD i mod 8),
    // in the order the row blocks will want them.  A launch's weights are cold (the L2s are invalidated between launches, a pass streams 1.7 GB through
    // them): without this every k-tile of every row block waits a memory round trip of ~2 000 cycles behind a request that is ~600 cycles old, and the
    // whole tail lasts as long as its seven launches did (profiles/r04_tblock_tail_v1_ab.txt)
    if ((int)blockIdx.x >= nblk) {
        extern __shared__ __attribute__((aligned(16))) char lds_pf[];
        char* dummy = lds_pf + __builtin_amdgcn_readfirstlane(tid >> 6) * 256;       // (256 bytes per wave)
        // PACED: a request for everything at once queues the row blocks' own first requests behind 26 000 lines per XCD (the first contractions then
        // wait longer than without any prefetch); the weights are asked for roughly at the rate the row blocks consume them, a few microseconds ahead
        auto nap = [&](int n) __attribute__((always_inline)) { for (; n > 0; n -= 127) __builtin_amdgcn_s_sleep(127); };
        touch_lines(p.wq2, 1, C * C * 2, 0, tid, dummy);        // (to_out1's own weights: the row blocks request them at entry themselves)
        {
            constexpr int DP = (D + 15) / 16 * 16;
            const int imgs = p.M / p.rows_per_img, bytes = imgs * p.heads * TKT * 16 * DP * 2;
            touch_lines(p.kp, 1, bytes, 0, tid, dummy);
            touch_lines(p.vtp, 1, bytes, 0, tid, dummy);
        }
        touch_lines(p.wo2, 1, C * C * 2, 0, tid, dummy);
        nap(p.pf_sleep0);
        for (int c = 0; c < NCHUNK; c++) {
            touch_lines((const char*)p.w1 + c * CH1, C / 8, HC * 16, 2 * F * 16, tid, dummy);              // value rows of chunk c, every k chunk
            touch_lines((const char*)p.w1 + GATE + c * CH1, C / 8, HC * 16, 2 * F * 16, tid, dummy);       // gate rows
            touch_lines((const char*)p.w2 + c * CH2, 1, CH2, 0, tid, dummy);                                // ff.net.2 over this chunk's hidden columns
            if (c == NCHUNK / 2 && p.wpo) touch_lines(p.wpo, 1, C * C * 2, 0, tid, dummy);
            nap(p.pf_sleep);
        }
        return;
    }

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const X = lds;                       // the residual stream of the row block (x0 -> x1 -> x2)
    char* const P = lds + IMG;                 // the current A operand (a1 -> LN(x1) -> a2 -> LN(x2) -> x3)
    char* const R = lds + 2 * IMG;             // q; then the two GEGLU chunk buffers
    // the block's small operands: nine [C] vectors and the [2F] bias of ff.net.0.proj, copied once (zeros where an op has no bias)
    char* const VEC = lds + 3 * IMG;
    enum { V_BO1, V_G2, V_BE2, V_BQ2, V_BO2, V_G3, V_BE3, V_B2, V_BPO, V_B1 };
    auto vec = [&](int k) __attribute__((always_inline)) { return VEC + k * (C * 2); };
    {
        const f16* src[10] = {p.bo1, p.g2, p.be2, p.bq2, p.bo2, p.g3, p.be3, p.b2, p.bpo, p.b1};
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const int n16 = (k == V_B1 ? 2 * F : C) / 8;
            for (int i = tid; i < n16; i += 256) {
                f16x8 v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (src[k]) v = *reinterpret_cast<const f16x8*>(src[k] + i * 8);
                *reinterpret_cast<f16x8*>(vec(k) + i * 16) = v;
            }
        }
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    const long row0 = (long)blockIdx.x * RB;
    const int img = (int)(row0 / p.rows_per_img);
    const int a_rd = l16 * 128 + ((g ^ (l16 & 7)) << 4);
    const int nb = wave * (C / 4);             // this wave's output columns of a C-wide contraction
    long long* const stamps = reinterpret_cast<long long*>(p.dbg[7]);      // dev probe: 16 wall-clock stamps (100 MHz) per row block, NULL in production
    auto stamp = [&](int k) __attribute__((always_inline)) { if (stamps && tid == 0) stamps[(long)blockIdx.x * 32 + k] = wall_clock64(); };
    stamp(0);

    f16x8 bq[NS][10];      // feed-forward tiles: NS - 1 ahead
    f16x8 bs[KT][10];      // to_out1's tiles: all of them, requested at entry (square_stage)
    // per-lane weight offsets (bytes; the bases stay in scalar registers): k chunk g, row nb + l16 (+ 16 j)
    const unsigned lo_c = (unsigned)((g * C + nb + l16) * 16);                   // [C/8][C][8] weights and ff.net.2 [F/8][C][8] (+ chunk * CH2)
    const unsigned lo_1 = (unsigned)((g * 2 * F + wave * 32 + l16) * 16);        // ff.net.0.proj [C/8][2F][8]: + chunk * CH1; gate rows GATE further
    const __amdgpu_buffer_rsrc_t u_o1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo1, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_q2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wq2, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_o2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo2, 0, C * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 2 * F * C * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, C * F * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_po = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpo, 0, p.wpo ? C * C * 2 : 0, 0x00020000);
    const bool has_po = p.wpo != nullptr;
    // requests by position: a [C x C] contraction's tile t; the feed-forward's tile `pos` of chunk iteration cc -- positions 0 .. KT-1 = ff.net.0.proj of
    // chunk cc (value + gate rows), KT .. PER-1 = ff.net.2 over chunk cc - 1
    auto ld_cc = [&](f16x8 (&b)[10], __amdgpu_buffer_rsrc_t rs, int t) __attribute__((always_inline)) { load_b<NTW>(b, rs, lo_c, t * KT_C, KS_C); };
    auto ld_ff = [&](f16x8 (&b)[10], int cc, int pos) __attribute__((always_inline)) {
        if (pos < KT) load_b_geglu(b, u_1, lo_1, cc * CH1 + pos * KT_1, KS_1, GATE);
        else load_b<NTW>(b, u_2, lo_c, (cc - 1) * CH2 + (pos - KT) * KT_C, KS_C);
    };

    static_for<KT>([&](auto tc) __attribute__((always_inline)) { ld_cc(bs[decltype(tc)::value], u_o1, decltype(tc)::value); });      // every tile of the first contraction: in flight while the row block arrives

    // ---- the row block: a1 -> P, x0 -> X (LDS-DMA, the swizzle on the source side as in gemm2_kernel) ----------------------------------
    {
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a1 + row0 * C), 0, RB * C * 2, 0x00020000);
        __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x0 + row0 * C), 0, RB * C * 2, 0x00020000);
        const int rsub = lane >> 3, gch = (lane & 7) ^ rsub;
#pragma unroll
        for (int kt = 0; kt < KT; kt++)
#pragma unroll
            for (int qq = 0; qq < RT / 2; qq++) {
                const int q8 = qq * 4 + wave;                               // 8-row group inside the tile
                const unsigned off = (unsigned)(((q8 * 8 + rsub) * C + kt * 64 + gch * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(P + kt * TB + q8 * 1024), 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(X + kt * TB + q8 * 1024), 16, off, 0, 0, 0);
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    stamp(1);

    // ---- x1 = to_out(a1) + x0 ---------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[RT][NTW];
        zero_acc(acc);
        square_stage<RT, KT, NTW>(P, a_rd, bs, acc, [&](f16x8 (&)[10], int t) __attribute__((always_inline)) { if (t < AH) ld_cc(bq[t % NS], u_q2, t); },
                              [&](int t) __attribute__((always_inline)) { stamp(16 + t); });
        stamp(21);
        epi_to_lds<RT, NTW, true>(acc, vec(V_BO1), nb, X, X, lane);
        stamp(22);
    }
    lds_barrier();
    stamp(2);
    dump_img<RT, C>(X, p.dbg[0], row0, tid);
    ln_rows<RT, C>(X, P, vec(V_G2), vec(V_BE2), p.eps2, tid);
    lds_barrier();
    stamp(3);
    dump_img<RT, C>(P, p.dbg[1], row0, tid);

    // ---- q = to_q(LN(x1)) ---------------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[RT][NTW];
        zero_acc(acc);
        gemm_stage<RT, KT, NTW, 0, NS>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { ld_cc(b, u_q2, t); },
                                   [&](f16x8 (&b)[10], int j) __attribute__((always_inline)) { ld_cc(b, u_o2, j); });
        epi_to_lds<RT, NTW, false>(acc, vec(V_BQ2), nb, nullptr, R, lane);
    }
    lds_barrier();
    stamp(4);
    dump_img<RT, C>(R, p.dbg[2], row0, tid);

    // ---- a2 = cross-attention(q, K, V): heads split over the waves -------------------------------------------------------------------
    {
        constexpr int DS = (D + 15) / 16, DP = DS * 16, TKP = TKT * 16;
        const int hpw = p.heads >> 2;
        const long per_img = (long)p.heads * TKP * DP;
        cross_attention<RT, D, TKT>(R, P, p.kp + img * per_img, p.vtp + img * per_img, wave * hpw, hpw, p.sc_log2e, p.Tk, lane);
    }
    lds_barrier();
    stamp(5);
    dump_img<RT, C>(P, p.dbg[3], row0, tid);

    // ---- x2 = to_out(a2) + x1 -----------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[RT][NTW];
        zero_acc(acc);
        gemm_stage<RT, KT, NTW, KT, NS>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { ld_cc(b, u_o2, t); },
                                    [&](f16x8 (&b)[10], int j) __attribute__((always_inline)) { ld_ff(b, 0, j); });
        epi_to_lds<RT, NTW, true>(acc, vec(V_BO2), nb, X, X, lane);
    }
    lds_barrier();
    stamp(6);
    dump_img<RT, C>(X, p.dbg[4], row0, tid);
    ln_rows<RT, C>(X, P, vec(V_G3), vec(V_BE3), p.eps3, tid);
    lds_barrier();
    stamp(7);
    dump_img<RT, C>(P, p.dbg[5], row0, tid);

    // ---- x3 = ff.net.2(GEGLU(ff.net.0.proj(LN(x2)))) + x2, the hidden activation 128 columns at a time ----------------------------------
    // k-tile sequence: G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(NCHUNK-1) G2(NCHUNK-2) | G2(NCHUNK-1); G1 = KT tiles (4 fragments each), G2 = G2T tiles (NTW)
    f32x4 accY[RT][NTW];
    zero_acc(accY);
    {
        constexpr int S0 = 2 * KT;                 // sequence index (slot = index % NS) of G1(0)'s first k-tile: to_q and to_out2 came first (to_out1 has slots of its own)
        f32x4 accG[RT][4];
        // GEGLU epilogue of chunk c into hidden buffer c & 1 (value tiles 0, 1; gate tiles 2, 3)
        auto geglu_store = [&](int c) __attribute__((always_inline)) {
            char* H = R + (c & 1) * (G2T * TB);
#pragma unroll
            for (int jt = 0; jt < 2; jt++) {
                const int hc = wave * 32 + jt * 16 + g * 4;          // column inside the chunk
                const f16x4 bv = *reinterpret_cast<const f16x4*>(vec(V_B1) + (c * HC + hc) * 2);
                const f16x4 bg = *reinterpret_cast<const f16x4*>(vec(V_B1) + (F + c * HC + hc) * 2);
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    f16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (f16)((accG[i][jt][r] + (float)bv[r]) * osg_gelu_erf(accG[i][2 + jt][r] + (float)bg[r]));
                    *reinterpret_cast<f16x4*>(H + img_off<RT>(i * 16 + l16, hc)) = o;
                }
            }
        };
        // chunk 0: projection only (its last AH requests are the first tiles of iteration 1)
        zero_acc(accG);
        gemm_stage<RT, KT, 4, S0, NS>(P, a_rd, bq, accG, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { ld_ff(b, 0, t); },
                                  [&](f16x8 (&b)[10], int j) __attribute__((always_inline)) { ld_ff(b, 1, j); });
        geglu_store(0);
        lds_barrier();
        stamp(8);
        // iteration c >= 1: projection of chunk c, then ff.net.2 over chunk c - 1 beside the GEGLU arithmetic of chunk c.  Its tile `pos` sits in slot
        // (ST + pos) % NS; the request that goes out with it is position pos + AH of this iteration, or of the next one (the last iteration is followed by
        // positions KT .. PER-1 of a pseudo-iteration NCHUNK -- ff.net.2 over the last chunk -- and then by proj_out)
        auto ff_iter = [&](auto st, int c) __attribute__((always_inline)) {
            constexpr int ST = decltype(st)::value;
            const bool last = c == NCHUNK - 1;
            auto request = [&](f16x8 (&b)[10], int pos) __attribute__((always_inline)) {     // pos: compile-time after unrolling
                if (pos < PER) ld_ff(b, c, pos);
                else if (!last) ld_ff(b, c + 1, pos - PER);
                else ld_ff(b, NCHUNK, KT + pos - PER);
            };
            zero_acc(accG);
            static_for<KT>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                request(bq[(ST + t + AH) % NS], t + AH);
                __builtin_amdgcn_sched_barrier(0);
                mma_ktile<RT, 4>(P + t * TB, a_rd, bq[(ST + t) % NS], accG);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (c == 3) stamp(24); else if (c == 4) stamp(28);
            // ff.net.2 over chunk c - 1, then the GEGLU arithmetic of chunk c.  (Slicing that arithmetic -- ~180 VALU / transcendental instructions per tile
            // pair, independent of these MFMAs -- between the groups of four MFMAs was measured SLOWER: 2.72 instead of 0.93 + 1.18 us per chunk,
            // profiles/r04_tblock_tail_stage_stamps_v5_geglu_interleaved_slower.txt -- one wave per SIMD issues in order, and the packed-f32 arithmetic takes the issue
            // slots the MFMAs need; sched_group_barrier patterns were not honoured at all)
            const char* Hp = R + ((c - 1) & 1) * (G2T * TB);
            static_for<G2T>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                request(bq[(ST + KT + u + AH) % NS], KT + u + AH);
                __builtin_amdgcn_sched_barrier(0);
                mma_ktile<RT, NTW>(Hp + u * TB, a_rd, bq[(ST + KT + u) % NS], accY);
            });
            if (c == 3) stamp(25); else if (c == 4) stamp(29);
            geglu_store(c);
            if (c == 3) stamp(26); else if (c == 4) stamp(30);
            lds_barrier();
            if (c == 2) stamp(23); else if (c == 3) stamp(27); else if (c == 4) stamp(31);
        };
        constexpr int B0 = (S0 + KT) % NS;          // slot of the first tile of iteration 1; iteration c starts at (B0 + (c - 1) PER) % NS = (B0 + c - 1) % NS
        int c = 1;
        for (; c + NS <= NCHUNK; c += NS) {
            ff_iter(std::integral_constant<int, B0 % NS>{}, c);
            ff_iter(std::integral_constant<int, (B0 + 1) % NS>{}, c + 1);
            if constexpr (NS == 3) ff_iter(std::integral_constant<int, (B0 + 2) % NS>{}, c + 2);
        }
        constexpr int REM = (NCHUNK - 1) % NS;      // iterations left over (their slots continue the rotation from B0)
        if constexpr (REM >= 1) ff_iter(std::integral_constant<int, B0 % NS>{}, c);
        if constexpr (REM >= 2) ff_iter(std::integral_constant<int, (B0 + 1) % NS>{}, c + 1);
        stamp(9);
        // ff.net.2 over the last chunk (positions KT .. PER-1 of pseudo-iteration NCHUNK); the requests that go out with it are proj_out's first tiles
        {
            constexpr int ST = (S0 + KT + (NCHUNK - 1) * PER) % NS;        // slot of its first tile
            const char* Hp = R + ((NCHUNK - 1) & 1) * (G2T * TB);
            static_for<G2T>([&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                if constexpr (u + AH < G2T) ld_ff(bq[(ST + u + AH) % NS], NCHUNK, KT + u + AH);
                else if (has_po) ld_cc(bq[(ST + u + AH) % NS], u_po, u + AH - G2T);
                __builtin_amdgcn_sched_barrier(0);
                mma_ktile<RT, NTW>(Hp + u * TB, a_rd, bq[(ST + u) % NS], accY);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    stamp(10);
    f16x4 rg[RT][NTW];                          // proj_out's residual x_in: requested now, behind the last weight tiles, consumed after the contraction
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int j = 0; j < NTW; j++) rg[i][j] = f16x4{0, 0, 0, 0};
    if (!has_po) {
        if (p.out2) epi_to_global<RT, NTW, true, true>(accY, vec(V_B2), nb, X, rg, p.out, p.ldo, p.out2, p.ldo2, row0, lane);
        else epi_to_global<RT, NTW, true, false>(accY, vec(V_B2), nb, X, rg, p.out, p.ldo, nullptr, 0, row0, lane);
        if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(13); }
        return;
    }
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int j = 0; j < NTW; j++) rg[i][j] = *reinterpret_cast<const f16x4*>(p.xin + (row0 + i * 16 + l16) * C + nb + j * 16 + g * 4);
    // x3 into P (every wave is past its last read of LN(x2): the barrier of the last chunk), then y = proj_out(x3) + x_in
    epi_to_lds<RT, NTW, true>(accY, vec(V_B2), nb, X, P, lane);
    lds_barrier();
    stamp(11);
    dump_img<RT, C>(P, p.dbg[6], row0, tid);
    {
        f32x4 acc[RT][NTW];
        zero_acc(acc);
        constexpr int SS = 2 * KT + KT + (NCHUNK - 1) * PER + G2T;
        gemm_stage<RT, KT, NTW, SS, NS>(P, a_rd, bq, acc, [&](f16x8 (&b)[10], int t) __attribute__((always_inline)) { ld_cc(b, u_po, t); },
                                    [&](f16x8 (&)[10], int) __attribute__((always_inline)) {});
        stamp(12);
        if (p.out2) epi_to_global<RT, NTW, false, true>(acc, vec(V_BPO), nb, nullptr, rg, p.out, p.ldo, p.out2, p.ldo2, row0, lane);
        else epi_to_global<RT, NTW, false, false>(acc, vec(V_BPO), nb, nullptr, rg, p.out, p.ldo, nullptr, 0, row0, lane);
        if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(13); }
    }
}

// K / V of a cross-attention re-packed for cross_attention(): rows [imgs][Tk] of ONE matrix `base` (row pitch ld), job j's K at column k_col, its V at
// column v_col (head h a further h D columns in) -> kp [img][head][TKP][DP] at dst + job.dst, vtp [img][head][DP][TKP] right behind it; zero padding
struct PackJob { int k_col, v_col, D, dst; };
__global__ __launch_bounds__(256) void kv_pack_kernel(const f16* __restrict__ base, long ld, int Tk, int heads, int imgs, const PackJob* __restrict__ jobs, f16* __restrict__ dst) {
    const int per_job = imgs * heads;
    const int j = blockIdx.x / per_job, ih = blockIdx.x - j * per_job, img = ih / heads, h = ih - img * heads;
    const PackJob job = jobs[j];
    const int D = job.D, DP = (D + 15) / 16 * 16, TKP = 80, NC = DP / 8;      // (D % 8 == 0: a 16-byte chunk is all inside or all outside the head)
    f16* kp = dst + job.dst + (long)ih * TKP * DP;
    f16* vtp = dst + job.dst + (long)per_job * TKP * DP + (long)ih * TKP * DP;
    const f16* k = base + job.k_col + h * D + (long)img * Tk * ld;
    const f16* v = base + job.v_col + h * D + (long)img * Tk * ld;
    for (int idx = threadIdx.x; idx < TKP * NC; idx += 256) {
        const int t = idx / NC, d0 = (idx - t * NC) * 8;
        const bool in = t < Tk && d0 < D;
        f16x8 kv = f16x8{0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (in) {
            kv = *reinterpret_cast<const f16x8*>(k + (long)t * ld + d0);
            vv = *reinterpret_cast<const f16x8*>(v + (long)t * ld + d0);
        }
        *reinterpret_cast<f16x8*>(kp + t * DP + d0) = kv;
#pragma unroll
        for (int e = 0; e < 8; e++) vtp[(d0 + e) * TKP + t] = vv[e];
    }
}

// [N][K] (k contiguous) -> kn8 [K/8][N][8]: one 16-byte chunk per thread
__global__ __launch_bounds__(256) void kn8_pack_kernel(const f16x8* __restrict__ src, f16x8* __restrict__ dst, int N, int K8) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * K8) return;
    const int kc = (int)(idx / N), n = (int)(idx - (long)kc * N);
    dst[idx] = src[(long)n * K8 + kc];
}


}   // namespace osg_tb

extern "C" {

int osg_tblock_tail_supported(int M, int rows_per_img, int C, int heads, int Tk) {
    return C == 320 && heads == 8 && M > 0 && M % 32 == 0 && rows_per_img % 32 == 0 && Tk >= 1 && Tk <= 80;
}

size_t osg_tblock_kv_pack_elems(int imgs, int heads, int D) { return (size_t)imgs * heads * 80 * (size_t)((D + 15) / 16 * 16); }

int osg_tblock_pack_weight(osg_ctx* ctx, const void* w_nk, int N, int K, void* w_kn8) {
    if (N < 1 || K < 8 || K % 8) OSG_FAIL(ctx, "osg_tblock_pack_weight: K must be a multiple of 8");
    const long n = (long)N * (K / 8);
    hipLaunchKernelGGL(osg_tb::kn8_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->compute, (const f16x8*)w_nk, (f16x8*)w_kn8, N, K / 8);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_tblock_kv_pack_jobs(osg_ctx* ctx, const void* base, long ld, int imgs, int Tk, int heads, int njobs, const int* jobs_dev, void* dst) {
    if (Tk < 1 || Tk > 80 || njobs < 1 || imgs < 1 || heads < 1 || ld % 8) OSG_FAIL(ctx, "osg_tblock_kv_pack_jobs: unsupported shape (1 <= Tk <= 80, row pitch a multiple of 8 elements; columns and head dims multiples of 8)");
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
    static const int pfw = getenv("OSG_TBLOCK_PREFETCH") ? atoi(getenv("OSG_TBLOCK_PREFETCH")) : 8;      // dev knobs: number of weight-prefetching workgroups (0 = none),
    static const int pfs0 = getenv("OSG_TBLOCK_PF_SLEEP0") ? atoi(getenv("OSG_TBLOCK_PF_SLEEP0")) : 100;  // their pauses, x 64 cycles: behind the square weights ...
    static const int pfs = getenv("OSG_TBLOCK_PF_SLEEP") ? atoi(getenv("OSG_TBLOCK_PF_SLEEP")) : 60;      // ... and behind every feed-forward chunk
    p.pf_sleep0 = pfs0; p.pf_sleep = pfs;
    // Rows per block: 64 rows give every weight fragment four row tiles of MFMA work, but M = 8 192 rows (SD 1.5 at 64x64, cond + uncond) are then 128 workgroups
    // on 256 CUs; 32-row blocks put one workgroup on every CU at twice the weight traffic out of the L2s.  OSG_TBLOCK_ROWS = 32 | 64 overrides the choice.
    // (Two register slots for the feed-forward's tiles: three -- two tiles ahead -- measured no faster, profiles/r04_tblock_tail_probe_v3.txt, and with the
    // square contractions' five slots beside them hipcc 7.2 crashes in its 'Rewrite AGPR-Copy-MFMA' pass.)
    static const int rows_env = getenv("OSG_TBLOCK_ROWS") ? atoi(getenv("OSG_TBLOCK_ROWS")) : 0;
    int rows = a->rows_per_block == 32 || a->rows_per_block == 64 ? a->rows_per_block : rows_env == 32 || rows_env == 64 ? rows_env : (a->M / 64 < 2 * ctx->num_cu ? 32 : 64);
    if (a->rows_per_block != 0 && a->rows_per_block != 32 && a->rows_per_block != 64) OSG_FAIL(ctx, "osg_tblock_tail: rows_per_block must be 0, 32 or 64");
    if (a->M % 64 || a->rows_per_img % 64) {
        if (a->rows_per_block == 64) OSG_FAIL(ctx, "osg_tblock_tail: 64-row blocks need M and rows_per_img to be multiples of 64");
        rows = 32;
    }
    const int nblk = a->M / rows;
    // eight more workgroups pull the weights into the eight L2s -- where the launch leaves CUs free for them (64-row blocks at M = 8 192); beside one row block per CU
    // they cost more than they bring (32-row blocks, 256 + 8 workgroups: 76.7 against 71.2 us per launch, profiles/r04_tblock_tail_rows_probe.txt)
    const int npf = nblk >= 64 && nblk + pfw <= ctx->num_cu ? pfw : 0;
    auto launch = [&](auto kern, int smem, unsigned long long& attr_mask) -> int {
        if (osg_first_on_device(attr_mask)) {
            OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(nblk + npf)), dim3(256), smem, ctx->compute, p);
        return 0;
    };
    constexpr int vec_bytes = (9 * 320 + 8 * 320) * 2;     // the small operands behind the three row-block images
    static unsigned long long attr64 = 0, attr32 = 0, attr32n3 = 0;   // (per-device memos, osg_common.h osg_first_on_device)
    static const bool ns3 = getenv("OSG_TBLOCK_NS") && atoi(getenv("OSG_TBLOCK_NS")) == 3;     // dev knob: two weight tiles ahead (32-row blocks only)
    int rc;
    if (rows == 64) rc = launch(osg_tb::tblock_tail_kernel<4, 320, 40, 5, 2>, 3 * 5 * osg_tb::kTileBytes<4> + vec_bytes, attr64);
    else if (ns3) rc = launch(osg_tb::tblock_tail_kernel<2, 320, 40, 5, 3>, 3 * 5 * osg_tb::kTileBytes<2> + vec_bytes, attr32n3);
    else rc = launch(osg_tb::tblock_tail_kernel<2, 320, 40, 5, 2>, 3 * 5 * osg_tb::kTileBytes<2> + vec_bytes, attr32);
    if (rc) return rc;
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}   // extern "C"
