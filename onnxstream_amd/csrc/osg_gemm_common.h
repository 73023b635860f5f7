// Shared by the contraction kernels of libosgpu (osg_gemm.hip, osg_conv3x3.hip): launch parameters and the fused epilogue.
#pragma once
#include <utility>
#include <vector>
#include "osg_common.h"

#include <type_traits>
#ifndef OSG_EPI_STORE_AUX
#define OSG_EPI_STORE_AUX 0     // < 0: plain pointer stores with exec masks (rounds 1-5); 0 / 16: buffer stores, plain / sc1 write-through (A/B builds: profiles/r06_epilogue_store_ab.txt)
#endif
namespace osg_mm {

// GroupNorm statistics from the PRODUCER's epilogue (round 3).  A convolution whose output a GroupNorm reads adds, per (image, group), the sum and the sum of
// squares of the f16 values it stores to a table the normalisation then only has to read: int64 FIXED-POINT sums (kStatSX / stat_q_scale fractional bits) --
// integer additions commute, so the atomics of the workgroups give the same bits whatever order they arrive in (fp32 atomics would make the pass
// non-deterministic).  A wave reduces its tile's columns over its rows (DPP butterfly over the 16 lanes that hold one column quad), stages the channel sums in
// LDS, and one lane per GROUP the wave's columns touch adds them up and issues the two atomics.
// The table has kStatCopies = 8 copies, [copy][image][groups][2]: a workgroup adds to the copy of the XCD it runs on (XCC_ID) with an atomic that is executed IN
// THAT XCD's L2 (workgroup scope: no sc1) -- device-scope atomics on one address go to the memory side and serialise at ~100 ns each: 70 arrivals per cell made
// every producing launch 9 us longer (profiles/r03_gn_stats_ab.txt).  All workgroups that share a copy share the L2 that executes their atomics; the copies
// are written back at the end of the kernel like any other store, and the reader sums the eight.  per_xcd = 0 (a device without eight XCDs 0..7): copy 0
// only, device-scope atomics.
// cpg / ch_off: channels per group and the channel offset of this launch's output inside the tensor the GroupNorm normalises (a Concat slot); two sinks:
// the output itself and its second destination (C2).
struct StatSink {
    long long* table;
    int groups, cpg, ch_off;
};
constexpr int kStatCopies = 8;
constexpr float kStatSX = 1048576.f;   // 2^20: |sum x| of a group up to 2^43
// The sum of squares is scaled by 2^e with e chosen from the group's size (elements per image and group = rows x channels per group; writers and readers derive it
// from the same two numbers): e = 36 - ceil(log2(elements)), clamped to [10, 20].  Round 4 used 2^20 for every tensor: the int64 sum wrapped at sum x^2 >= 2^43,
// i.e. at an rms of 1 450 over the 2^22 elements of the VAE decoder's largest groups, silently (advisor, round 4); 2^8 before that rounded the partial sums of a
// small-magnitude tensor (|x| ~ 1e-3) to zero (advisor, round 3).  With the size-dependent scale the sum holds any group whose rms stays below ~11 500 (2^27 / 2^e
// per element on average; f16 itself ends at 65 504), and the rounding of the per-wave partials -- at most half a unit each -- averages out over the
// elements / 256 partials of a group: a group of 2^22 elements of |x| ~ 1e-3 keeps its variance to ~0.1 %.
__host__ __device__ inline int stat_q_shift(long elems) {
    int lg = 0;
    while ((1L << lg) < elems) lg++;
    const int e = 36 - lg;
    return e < 10 ? 10 : (e > 20 ? 20 : e);
}
__host__ __device__ inline float stat_q_scale(long elems) { return (float)(1L << stat_q_shift(elems)); }

struct GemmParams {
    // ---- round 6: field ORDER = the order a workgroup needs them.  A launch of the pass lasts as long as one workgroup, and a workgroup's first DMA request waited for
    // ~6 dependent round trips of scalar loads from the kernel-argument segment (hipcc loads a field where it is first used); the kernels now pull the first group in
    // ONE batch at entry (OSG_PIN in gemm2_kernel), the second group (convolution geometry) and the third (operands of the epilogue prefetch) right behind it.
    // group 1: tile mapping + the DMA addressing of both operands
    const f16* A;
    const f16* Bt;
    long long* kdbg;             // developer probe (OSG_KDBG=1, tools/kernel_phase_probe.py): per-workgroup phase timestamps, 8 x int64 per workgroup (s_memrealtime, 100 MHz); NULL in normal operation
    long lda;
    long strideA, strideB;
    int M, N, K;
    int splits, k_per_split;
    unsigned a_bytes, b_bytes;   // (v2 kernels) buffer-descriptor extents of one batch item of A / Bt
    int mt, nt, n_major;         // tile grid and the order tiles are walked inside an XCD's contiguous chunk
    int grid;                    // gemm2_kernel: workgroups of the launch (gridDim.x without the trip to the hidden arguments)
    // group 2: conv geometry (CONV only)
    int H, W, Cin, Ho, Wo, KW, sh, sw, pt, pl;
    // group 3: what the epilogue prefetch reads before the first tile is requested
    const void* bias;
    const f16* residual;
    const f16* rowbias;          // optional [M / rb_rows][rb_ld] per-image channel bias (the resnet time-embedding add)
    long rb_ld;
    long strideC;
    const float* ln_c1;          // osg_gemm_ln: LayerNorm over K folded into this GEMM -- c1[n] = sum_k W'[n][k]; bias holds c2 (f32)
    const float* rs_in;          //   row statistics of A emitted by the GEMM that produced it: [M][K/32][2] (sum, sum of squares) per 32 columns
    int rb_rows;
    int bias_f32, act;
    int no_epre;                 // OSG_NO_EPI_PREFETCH=1 (A/B): the epilogue fetches its operands on demand, as before round 3
    int rs_np;                   //   = N / 32
    // the rest: epilogue / split-K
    f16* C;
    float* partial;
    long a_bytes_l;              // conv: byte size of the whole NHWC input (host side, before the 2 GiB check)
    int* tickets;                // split-K arrival / publication counters (two per output tile, splitk_fold_acc), zero between launches
    int* xcd_err;                // host-mapped flag the bounded wait of splitk_fold_acc raises (osg_ctx; checked by osg_sync / osg_download)
    float w_scale;               // W8 (gemm2_kernel / conv3x3_kernel with WQ = 1, osg_gemm_w8.hip): Bt holds uint8 codes [N][K], w = (q - w_zp) * w_scale
    int w_zp;
    const float* wq_sc;          //   per-output-column scale / zero point [N] (merged projections: one pair per member); NULL = the scalars above
    const float* wq_zp;
    int w8;                      //   host side: the launch takes the WQ = 1 instantiations
    float ln_eps;
    float* rs_out;               // osg_gemm_rowstats: this GEMM's epilogue also emits [M][rs_np][2] partial row statistics of its f16 output
    // output VIEWS (round 3: skip tensors written straight into their Concat slot, no copy launch): C rows are `ldc` elements apart (0 = dense, N), and
    // the finished f16 values are stored a second time to C2 (rows ldc2 apart) when it is set -- the dense tensor for the layers that read it as it is,
    // the column slice of the concatenated buffer for the up-block that reads the concatenation.  batch (strideC) launches take no views.
    long ldc;
    f16* C2;
    long ldc2;
    StatSink sink[2];            // (see StatSink) [0]: of C, [1]: of C2; table NULL = none
    int sink_hw;                 // output rows per image (a multiple of the tile height: a wave's rows lie in one image)
    int sink_imgs, sink_per_xcd; // images of the pass (the stride between table copies = sink_imgs * groups * 2), see StatSink
    // round 5: split-K folded by the LAST workgroup to arrive at a tile, in ACCUMULATOR layout (splitk_fold_acc below): 1 = on.  partial then holds
    // [tile][slice][TM * TN][256] f32x4 (a lane's accumulator tile = one 16-byte element: 1-KiB bursts per wave-instruction), tickets two words per tile.
    int fold_acc;
};
// a kernel-argument field pulled into a scalar register NOW: see GemmParams.  (An INPUT of an empty asm: an in-out operand would make the value opaque -- pointers lose
// their address space and every load through them becomes a flat_load, which counts on lgkmcnt AND vmcnt and breaks the counted waits of the k loop.)
#define OSG_PIN(x) asm volatile("" ::"s"(x))
// block placement: code a workgroup walks ONCE streams into the CU at ~3 ns per instruction (tools/floor_probe4) and every taken branch over a block it skips restarts the
// sequential fetch -- the rare side of the epilogue's uniform decisions goes out of line, the common path falls through (round 6: one such hint moved the pass by 0.03 ms)
#define OSG_LIKELY(x) __builtin_expect(!!(x), 1)
#define OSG_UNLIKELY(x) __builtin_expect(!!(x), 0)
// (h = false: no hint -- the 512- / 768-thread kernels, whose 256 / 168 registers per lane spill under a different block order)
#define OSG_UNLIKELY_IF(h, x) ((h) ? __builtin_expect(!!(x), 0) : !!(x))

// ---- W8A16: uint8 weight codes resident in HBM, dequantised between the LDS tile and the MFMA (round 6) ----------------------------------------------------
// The [BN][64] weight tile travels HBM -> L2 -> LDS as CODES (half the bytes of the f16 tile on every hop, and half the bytes through the LDS port the k loop is
// bound by); a lane reads the 8 codes of its fragment with one ds_read_b64 and turns them into the 8 halves the MFMA takes with 8 VALU operations:
//   v_perm_b32 pairs each code with the byte 0x64: the half 0x64cc IS 1024 + c exactly (ulp(1024) = 1 in binary16);
//   v_pk_add_f16 subtracts 1024 + zero_point: (q - zp), an integer of magnitude <= 255, exact.
// The MFMA therefore accumulates sum_k a[m][k] * (q[n][k] - zp[n]) in f32 and the tile's accumulators are multiplied by scale[n] ONCE, in f32, before the
// epilogue (w8_scale_acc).  The reference dequantises when it loads the weight, w = f16((float)(q - zp) * scale) (src/onnxstream.cpp:2887-2891, :3353), i.e. it
// rounds every weight to f16 first; this path does not round the weights at all: per output it differs from the reference's f16 GEMM on the dequantised
// weights by the rounding the reference applies to each weight (2^-12 relative per term, independent signs) and is the closer of the two to the f32 result.
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x8 w8_frag(u32x2v c, f16x2w zz) {
    union { unsigned u; f16x2w h; } d0, d1, d2, d3;
    d0.u = __builtin_amdgcn_perm(0x64646464u, c[0], 0x04010400u);
    d1.u = __builtin_amdgcn_perm(0x64646464u, c[0], 0x04030402u);
    d2.u = __builtin_amdgcn_perm(0x64646464u, c[1], 0x04010400u);
    d3.u = __builtin_amdgcn_perm(0x64646464u, c[1], 0x04030402u);
    d0.h -= zz; d1.h -= zz; d2.h -= zz; d3.h -= zz;
    return f16x8{d0.h[0], d0.h[1], d1.h[0], d1.h[1], d2.h[0], d2.h[1], d3.h[0], d3.h[1]};
}
// what a lane needs of the quantisation parameters, requested BEFORE the first tile (older than every tile load in the wave's in-order vector-memory queue: home by
// the time tile 0 is): zz[j] = 1024 + zero point of the weight row its B fragment j holds (row n0 + wn0 + 16 j + (lane & 15)), as a pair of halves; sc[j] = the scales
// of the 4 consecutive columns n0 + wn0 + 16 j + 4 (lane >> 4) .. + 3 its accumulator registers of column block j are.  Vectors (N % 4 == 0) or the scalars.
// SC = false (the tiles with no registers to spare): the scales are fetched on demand after the k loop (w8_scale_acc_late).
template <int TN, bool SC = true>
struct W8Ops {
    float zraw[TN];      // as loaded (converted by w8_finalize AFTER the first tiles are requested: a conversion on the spot would wait for the load there)
    f16x2w zz[TN];
    f32x4 sc[SC ? TN : 1];
};
template <int TN, bool SC>
__device__ __forceinline__ void w8_prefetch(const GemmParams& p, W8Ops<TN, SC>& w, int n0, int wn0, int lane) {
    if (p.wq_sc) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            w.zraw[j] = p.wq_zp[min(n0 + wn0 + j * 16 + (lane & 15), p.N - 1)];
            if constexpr (SC) w.sc[j] = *reinterpret_cast<const f32x4*>(p.wq_sc + min(n0 + wn0 + j * 16 + (lane >> 4) * 4, p.N - 4));
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            w.zraw[j] = (float)p.w_zp;
            if constexpr (SC) w.sc[j] = f32x4{p.w_scale, p.w_scale, p.w_scale, p.w_scale};
        }
    }
    asm volatile("" ::: "memory");
}
template <int TN, bool SC>
__device__ __forceinline__ void w8_finalize(W8Ops<TN, SC>& w) {
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const f16 h = (f16)(1024.0f + w.zraw[j]);
        w.zz[j] = f16x2w{h, h};
    }
}
// acc[m][n] *= scale[n], once, in f32, before the epilogue
template <int TM, int TN>
__device__ __forceinline__ void w8_scale_acc(const W8Ops<TN, true>& w, f32x4 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] *= w.sc[j];
}
template <int TM, int TN>
__device__ __forceinline__ void w8_scale_acc_late(const GemmParams& p, f32x4 (&acc)[TM][TN], int n0, int wn0, int lane) {
#pragma unroll
    for (int j = 0; j < TN; j++) {
        f32x4 s = {p.w_scale, p.w_scale, p.w_scale, p.w_scale};
        if (p.wq_sc) s = *reinterpret_cast<const f32x4*>(p.wq_sc + min(n0 + wn0 + j * 16 + (lane >> 4) * 4, p.N - 4));
#pragma unroll
        for (int i = 0; i < TM; i++) acc[i][j] *= s;
    }
}

__device__ __forceinline__ void kdbg_stamp(const GemmParams& p, int slot) {
    if (__builtin_expect(p.kdbg != nullptr, 0) && threadIdx.x == 0) p.kdbg[(long)blockIdx.x * 8 + slot] = wall_clock64();   // (out of line: see OSG_LIKELY)
}

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

// ---- epilogue operands fetched BEFORE the k loop (round 3) ----------------------------------------------------------------------------------------
// A launch of the UNet pass lasts as long as ONE workgroup does (~1 tile per CU), and a workgroup spent 2.5-4.8 us of its ~9-22 us in the epilogue
// (tools/kernel_phase_probe.py): bias, per-image bias and residual were requested only after the last MFMA -- a dependent trip to cold memory on the
// critical path of every launch.  They do not depend on the accumulators: the same loads are issued at kernel entry (unconditional, clamped
// addresses: a predicated load makes the compiler wait for it on the spot) and are home long before the k loop ends.  Values and arithmetic order in
// the epilogue are unchanged => bit-identical results.  Only for the fused epilogue of an unsplit launch with 4-aligned N.
// RB: the kernel can be handed a per-image bias (convolutions only) -- plain GEMMs do not spend registers on it.  ON = false: no prefetch at all (the
// 128 x 128 tiles, whose 16 accumulator tiles per wave leave no room for 16 more operand tiles: the old on-demand loads stay).
template <int TM, int TN, bool RB = true, bool ON = true>
struct EpiOps {
    bool have;
    f32x4 bias32[ON ? TN : 1];     // RAW as loaded (f32 or f16 bias: converted in the epilogue -- a conversion here would wait for the load on the spot)
    f16x4 bias16[ON ? TN : 1];
    f16x4 rb[ON && RB ? TM : 1][ON && RB ? TN : 1];
    f16x4 res[ON ? TM : 1][ON ? TN : 1];
};
template <int TM, int TN, bool RB, bool ON>
__device__ __forceinline__ void epi_prefetch(const GemmParams& p, EpiOps<TM, TN, RB, ON>& e, int m0, int n0, int wm0, int wn0, int lane, int zb) {
    const int N = p.N;
    e.have = ON && !p.no_epre && p.splits == 1 && (N & 3) == 0 && p.act != OSG_ACT_GEGLU && N >= 4 && (RB || !p.rowbias);
    if constexpr (!ON) return;
    if (__builtin_expect(!e.have, 0)) return;
    const f16* __restrict__ R = p.residual ? p.residual + zb * p.strideC : nullptr;
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = min(n0 + wn0 + j * 16 + (lane >> 4) * 4, N - 4);
        if (p.bias) {
            if (p.bias_f32) e.bias32[j] = *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
            else e.bias16[j] = *reinterpret_cast<const f16x4*>((const f16*)p.bias + n);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = min(m0 + wm0 + i * 16 + (lane & 15), p.M - 1);
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = min(n0 + wn0 + j * 16 + (lane >> 4) * 4, N - 4);
            if constexpr (RB) {
                if (p.rowbias) e.rb[i][j] = *reinterpret_cast<const f16x4*>(p.rowbias + (long)(m / p.rb_rows) * p.rb_ld + n);
            }
            if (R) e.res[i][j] = *reinterpret_cast<const f16x4*>(R + (long)m * N + n);
        }
    }
    asm volatile("" ::: "memory");
}

// ---- epilogue shared by both kernels: lane owns C[m][n..n+3], m = tile_m + (lane&15), n = tile_n + (lane>>4)*4 --------
// (operands are swapped -- weights feed the MFMA "A" port -- so the 4 accumulator registers of a lane are 4 consecutive
// output channels of one pixel); bias/residual/activation fused in f32 before the single RNE rounding to f16.
// The common case of the fused epilogue -- unsplit launch, N and the output pitches multiples of 4, no row statistics -- as COMPACT code (round 3): a
// launch lasts as long as one workgroup, every launch starts with a cold instruction cache, and the general epilogue below unrolls to ~10 000
// instructions for a 2 x 5 tile (tools/kernel_phase_probe.py: 4.2 us of a 22 us convolution were spent walking it).  Same loads, same additions in
// the same order, same rounding: identical bits.
// one (image, group) cell of a StatSink table += (S, Q): into the copy of this workgroup's XCD, executed in its L2 -- or copy 0, device scope
__device__ __forceinline__ void stat_add(int per_xcd, unsigned long long* table, long copy_stride, long cell, float S, float Q, float qscale) {
    const unsigned long long vs = (unsigned long long)__float2ll_rn(S * kStatSX), vq = (unsigned long long)__float2ll_rn(Q * qscale);
    if (per_xcd) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* c = table + (long)(xcc & 7u) * copy_stride + cell;
        __hip_atomic_fetch_add(c, vs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(c + 1, vq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        atomicAdd(table + cell, vs);
        atomicAdd(table + cell + 1, vq);
    }
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (quad butterfly, then the neighbouring quad, then the other half): every lane ends up with the row's total
__device__ __forceinline__ float dpp_sum16(float v) {
    v += dpp_get<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_get<0x124>(v);    // row_ror:4
    v += dpp_get<0x128>(v);    // row_ror:8
    return v;
}
// see StatSink.  o = the wave's finished f16 tile (row mb + 16 i, columns nb + 16 j + r), st = this wave's LDS staging area (2 * 16 TN * 4... floats:
// [column inside the wave][2]); mrow0 / ncol0 = first row / column of the wave's tile.
template <int TM, int TN>
__device__ __forceinline__ void gemm_colstats(const GemmParams& p, const f16x4 (&o)[TM][TN], int mrow0, int ncol0, int lane, float* st) {
    const int mb = mrow0 + (lane & 15);
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; i++)
                if (mb + i * 16 < p.M) {
                    const float f = (float)o[i][j][r];
                    s += f;
                    q = fmaf(f, f, q);
                }
            s = dpp_sum16(s);
            q = dpp_sum16(q);
            if ((lane & 15) == 0) {
                const int c = j * 16 + (lane >> 4) * 4 + r;
                st[c * 2] = s;
                st[c * 2 + 1] = q;
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (one wave: its own LDS writes are visible to its lanes once they have completed)
    const int n_img = mrow0 / p.sink_hw;
    const int wn = min(TN * 16, p.N - ncol0);              // valid columns of this wave
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const StatSink sk = p.sink[k];
        if (!sk.table || wn <= 0) continue;
        const int c_lo = ncol0 + sk.ch_off, c_hi = c_lo + wn;             // this wave's columns in the normalised tensor's channel numbering
        const int g = c_lo / sk.cpg + lane;
        if (g * sk.cpg < c_hi) {
            const int a = max(g * sk.cpg, c_lo) - c_lo, b = min((g + 1) * sk.cpg, c_hi) - c_lo;
            float S = 0.f, Q = 0.f;
            for (int c = a; c < b; c++) { S += st[c * 2]; Q += st[c * 2 + 1]; }
            stat_add(p.sink_per_xcd, reinterpret_cast<unsigned long long*>(sk.table), (long)p.sink_imgs * sk.groups * 2, ((long)n_img * sk.groups + g) * 2, S, Q,
                     stat_q_scale((long)p.sink_hw * sk.cpg));
        }
    }
}

template <int TM, int TN, bool RB, bool ON, bool BATCH = true>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane, int zb,
                                                   const EpiOps<TM, TN, RB, ON>& pre, float* stat_lds = nullptr) {
    // every uniform decision (which operands exist, which activation, a second destination) is taken ONCE, around a whole loop over the wave's tiles --
    // inside the unrolled loops the compiler would clone the tile code for every combination of them
    const int N = p.N;
    f16* __restrict__ C = p.C + zb * p.strideC;
    const f16* __restrict__ R = p.residual ? p.residual + zb * p.strideC : nullptr;
    const long ldc = p.ldc ? p.ldc : (long)N;
    const int nb = n0 + wn0 + (lane >> 4) * 4;
    const int mb = m0 + wm0 + (lane & 15);
    bool done = false;
    if constexpr (ON) {
        if (pre.have) {
            done = true;
            if (p.bias) {
                if (p.bias_f32) {
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++) acc[i][j] += pre.bias32[j];
                } else {
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[i][j][r] += (float)pre.bias16[j][r];
                }
            }
            if constexpr (RB) {
                if (p.rowbias) {
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
#pragma unroll
                            for (int r = 0; r < 4; r++) acc[i][j][r] += (float)pre.rb[i][j][r];
                }
            }
            if (R) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[i][j][r] += (float)pre.res[i][j][r];
            }
        }
    }
    if constexpr (BATCH) if (!done) {
        // operands on demand (the 128-row tiles: no registers to prefetch into before the k loop).  Round 6: EVERY load first -- unconditional, clamped, an absent operand
        // reads A instead and enters the sum as -0.0, the exact neutral element -- then the additions in the old order: loaded inside `if (p.bias) ... if (p.rowbias) ...
        // if (R) ...` each operand was waited for at the end of its block, three dependent trips to memory in a row on the critical path of the launch.  Same bits.
        // Through buffer descriptors: an absent operand gets an EMPTY descriptor (every load returns 0, no memory traffic), rows / columns outside the matrix are out of
        // range, and the TN blocks of a row are one address register + immediate offsets.
        const bool hb32 = p.bias && p.bias_f32, hb16 = p.bias && !p.bias_f32, hrb = p.rowbias != nullptr, hres = R != nullptr;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const unsigned nbytes = (unsigned)N * 2u;
        __amdgpu_buffer_rsrc_t rsB32 = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, hb32 ? nbytes * 2u : 0u, 0x00020000);
        __amdgpu_buffer_rsrc_t rsB16 = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, hb16 ? nbytes : 0u, 0x00020000);
        __amdgpu_buffer_rsrc_t rsRB = __builtin_amdgcn_make_buffer_rsrc((void*)p.rowbias, 0, hrb ? 0x80000000u : 0u, 0x00020000);
        __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, hres ? (unsigned)p.M * nbytes : 0u, 0x00020000);
        // Loads through an EMPTY descriptor are not free (tools/gemm_kloop_probe.py PROBE_NO_BIAS=1: the epilogue phase of a launch with NO operand at all took
        // 3.9 us on the 128 x 128 tile and 5.4 us on 128 x 160 against 1.95 / 2.44 us for the same stores behind the on-demand form): a launch with neither row
        // operand -- the merged q / k / v and to_q projections, the feed-forward outputs -- skips the 2 x TM x TN of them behind ONE uniform branch.  Same bits.
        // (the 20-block tiles -- 128 x 160 -- keep the one straight form: with a second variant of the epilogue beside it their 256 architectural registers spill)
        if (TM * TN <= 16 && OSG_LIKELY(!hrb && !hres)) {   // (the likely side falls through to the stores: a taken branch at the end of a launch is an instruction-cache miss on its critical path)
            if (hb32) {
                f32x4 b32[TN];
#pragma unroll
                for (int j = 0; j < TN; j++) b32[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB32, (unsigned)nb * 4u + j * 64, 0, 0));
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[i][j][r] += b32[j][r];
            } else if (hb16) {
                f16x4 b16[TN];
#pragma unroll
                for (int j = 0; j < TN; j++) b16[j] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rsB16, (unsigned)nb * 2u + j * 32, 0, 0));
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[i][j][r] += (float)b16[j][r];
            }
        } else {
        f32x4 b32[TN];
        f16x4 b16[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) {
            b32[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB32, (unsigned)nb * 4u + j * 64, 0, 0));
            b16[j] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rsB16, (unsigned)nb * 2u + j * 32, 0, 0));
        }
        // (tiles of more than 16 blocks per wave -- 128 x 160 -- take the row operands half a 16-row block at a time: 256 architectural registers)
        constexpr int IB = TM * TN <= 16 ? TM : 1, JB = (TM * TN <= 16 || TN % 2) ? TN : TN / 2;
        static_assert(TN % JB == 0, "column blocks per chunk");
#pragma unroll
        for (int i0 = 0; i0 < TM; i0 += IB)
#pragma unroll
            for (int j0 = 0; j0 < TN; j0 += JB) {
                f16x4 rbv[IB][JB], rsv[IB][JB];
#pragma unroll
                for (int ii = 0; ii < IB; ii++) {
                    const int m = mb + (i0 + ii) * 16;
                    const unsigned ro = hrb ? (unsigned)(((long)(min(m, p.M - 1) / (p.rb_rows > 0 ? p.rb_rows : 1)) * p.rb_ld + nb) * 2) : 0u;
                    const unsigned rs = (unsigned)m * nbytes + (unsigned)nb * 2u;
#pragma unroll
                    for (int jj = 0; jj < JB; jj++) {
                        rbv[ii][jj] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rsRB, ro + (j0 + jj) * 32, 0, 0));
                        rsv[ii][jj] = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rsR, rs + (j0 + jj) * 32, 0, 0));
                    }
                }
#pragma unroll
                for (int ii = 0; ii < IB; ii++)
#pragma unroll
                    for (int jj = 0; jj < JB; jj++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float x = acc[i0 + ii][j0 + jj][r];
                            x += p.bias ? (hb32 ? b32[j0 + jj][r] : (float)b16[j0 + jj][r]) : -0.0f;
                            x += hrb ? (float)rbv[ii][jj][r] : -0.0f;
                            x += hres ? (float)rsv[ii][jj][r] : -0.0f;
                            acc[i0 + ii][j0 + jj][r] = x;
                        }
            }
        }
    }
    if constexpr (!BATCH) if (!done) {   // (the 512- / 768-thread halo convolution: 256 / 168 registers per lane, no room for the batch) operands on demand (clamped addresses: the loads are unconditional, the stores below are not)
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = min(nb + j * 16, N - 4);
                f32x4 bv;
                if (p.bias_f32) bv = *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
                else {
                    const f16x4 b16 = *reinterpret_cast<const f16x4*>((const f16*)p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; r++) bv[r] = (float)b16[r];
                }
#pragma unroll
                for (int i = 0; i < TM; i++) acc[i][j] += bv;
            }
        }
        if (p.rowbias) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const long ro = (long)(min(mb + i * 16, p.M - 1) / p.rb_rows) * p.rb_ld;
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const f16x4 rb = *reinterpret_cast<const f16x4*>(p.rowbias + ro + min(nb + j * 16, N - 4));
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[i][j][r] += (float)rb[r];
                }
            }
        }
        if (R) {
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const long ro = (long)min(mb + i * 16, p.M - 1) * N;
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const f16x4 rv = *reinterpret_cast<const f16x4*>(R + ro + min(nb + j * 16, N - 4));
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[i][j][r] += (float)rv[r];
                }
            }
        }
    }
    if (OSG_UNLIKELY_IF(BATCH, p.act != OSG_ACT_NONE)) {
        const int act = p.act;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[i][j][r] = osg_apply_act(acc[i][j][r], act);
    }
    f16x4 o[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[i][j][r] = (f16)acc[i][j][r];
#if OSG_EPI_STORE_AUX >= 0
    // round 6: the finished tile through a buffer descriptor -- rows / columns outside the matrix get an out-of-range offset and the hardware drops them (no exec-mask
    // code per store), and the cache policy is a compile-time choice (OSG_EPI_STORE_AUX: 0 plain, 16 = sc1 write-through: nothing left dirty in the L2 for the
    // end-of-kernel write-back to wait for)
    {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, 0x80000000u, 0x00020000)   /* (outputs stay below 2 GiB: the planner's own limit on a tensor) */;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = mb + i * 16;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = nb + j * 16;
                const unsigned off = (m < p.M && n < N) ? (unsigned)(((long)m * ldc + n) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[i][j]), rsC, off, 0, OSG_EPI_STORE_AUX);
            }
        }
        if (OSG_UNLIKELY_IF(BATCH, p.C2 != nullptr)) {
            __amdgpu_buffer_rsrc_t rsC2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.C2, 0, 0x80000000u, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int m = mb + i * 16;
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int n = nb + j * 16;
                    const unsigned off = (m < p.M && n < N) ? (unsigned)(((long)m * p.ldc2 + n) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[i][j]), rsC2, off, 0, OSG_EPI_STORE_AUX);
                }
            }
        }
    }
#else
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = mb + i * 16;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = nb + j * 16;
            if (m < p.M && n < N) *reinterpret_cast<f16x4*>(C + (long)m * ldc + n) = o[i][j];
        }
    }
    if (p.C2) {
        f16* __restrict__ C2 = p.C2;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = mb + i * 16;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = nb + j * 16;
                if (m < p.M && n < N) *reinterpret_cast<f16x4*>(C2 + (long)m * p.ldc2 + n) = o[i][j];
            }
        }
    }
#endif
    if (OSG_UNLIKELY_IF(BATCH, stat_lds != nullptr)) gemm_colstats<TM, TN>(p, o, m0 + wm0, n0 + wn0, lane, stat_lds);
    if (OSG_UNLIKELY_IF(BATCH, p.rs_out != nullptr)) {
        // osg_gemm_rowstats: sums over this wave's 32-column slots of every row, of the ROUNDED outputs.  The four 16-lane groups of a row hold
        // different columns of the same slot pair: 2-step butterfly, one lane group stores
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = mb + i * 16;
            if (m >= p.M) continue;
#pragma unroll
            for (int h = 0; h < TN / 2; h++) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const int j = 2 * h + jj;
                    if (nb + j * 16 < N) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float f = (float)o[i][j][r];
                            s += f;
                            q = fmaf(f, f, q);
                        }
                    }
                }
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
}

template <int TM, int TN, bool RB, bool ON, bool BATCH = true>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane,
                                              int zb, int zslab, const EpiOps<TM, TN, RB, ON>& pre, float* stat_lds = nullptr) {
    const int N = p.N;
    if (OSG_LIKELY(p.splits == 1)) {
        if (OSG_LIKELY((N & 3) == 0 && ((p.ldc | p.ldc2) & 3) == 0)) {   // every 4-aligned shape: compact code (see gemm_epilogue_fast)
            gemm_epilogue_fast<TM, TN, RB, ON, BATCH>(p, acc, m0, n0, wm0, wn0, lane, zb, pre, stat_lds);
            return;
        }
        // ragged N (conv_out's 3 / 4 channels, odd test shapes): element by element
        f16* __restrict__ C = p.C + zb * p.strideC;
        const f16* __restrict__ R = p.residual ? p.residual + zb * p.strideC : nullptr;
        const long ldc = p.ldc ? p.ldc : (long)N;
        f16* __restrict__ C2 = p.C2;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int m = m0 + wm0 + i * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = n0 + wn0 + j * 16 + (lane >> 4) * 4;
                if (n >= N) continue;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (n + r >= N) break;
                    float x = acc[i][j][r];
                    if (p.bias) x += p.bias_f32 ? ((const float*)p.bias)[n + r] : (float)((const f16*)p.bias)[n + r];
                    if (p.rowbias) x += (float)p.rowbias[(long)(m / p.rb_rows) * p.rb_ld + n + r];
                    if (R) x += (float)R[(long)m * N + n + r];
                    const f16 o1 = (f16)osg_apply_act(x, p.act);
                    C[(long)m * ldc + n + r] = o1;
                    if (C2) C2[(long)m * p.ldc2 + n + r] = o1;
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


template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm0, int wn0, int lane, int zb, int zslab) {
    EpiOps<TM, TN, false, false> none;
    none.have = false;
    gemm_epilogue<TM, TN, false, false>(p, acc, m0, n0, wm0, wn0, lane, zb, zslab, none);
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
    // round 6: the bias of the wave's TN column blocks by ONE vector load each, all in flight together (rounds 1-5: eight 2-byte loads per output quad, on demand);
    // unconditional and clamped, an absent bias reads A and enters as -0.0 (the exact neutral element): same sums, same bits
    const bool hb32 = p.bias && p.bias_f32, hb16 = p.bias && !p.bias_f32;
    f32x4 b32[TN];
    f16x4 b16[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = min(n0 + wn0 + j * 16 + (lane >> 4) * 4, p.N - 4);
        b32[j] = *reinterpret_cast<const f32x4*>(hb32 ? (const float*)p.bias + n : (const float*)p.A);
        b16[j] = *reinterpret_cast<const f16x4*>(hb16 ? (const f16*)p.bias + n : p.A);
    }
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
                const float v = acc[i][j][r] + (p.bias ? (hb32 ? b32[j][r] : (float)b16[j][r]) : -0.0f);
                const float g = acc[i][j + 1][r] + (p.bias ? (hb32 ? b32[j + 1][r] : (float)b16[j + 1][r]) : -0.0f);
                o[r] = (f16)(v * osg_gelu_erf(g));
            }
            *reinterpret_cast<f16x4*>(C + (long)m * No + c) = o;
        }
    }
}

// ---- round 5: split-K without a reduce launch AND without a second pass over row-major slabs.  Every k-slice workgroup of a tile takes a ticket; all but the last
// publish their accumulators exactly as the lanes hold them ([TM * TN][256 threads] f32x4: each wave-instruction stores / loads 1 KiB contiguous) and leave;
// the last arriver waits for those publications (their authors HAVE arrived, so they are running and need nobody: the wait is bounded by their store drain),
// adds the slices in slice order with its own accumulators at its own position -- the same sum whoever comes last: bit-reproducible -- and runs the unchanged
// FUSED epilogue (bias, per-image bias, residual, activation, second destination) on the result.  Against the reduce launch: no launch boundary (~5 us in the
// captured pass), half the slab traffic at two slices, and no f32 row-major slab written and re-read through 64-byte pieces.  Slabs travel through memory
// (write-through stores, sc1 loads: the slices of a tile run on whatever XCDs the dispatcher picks), the counters are agent-scope atomics.  Returns true in the
// workgroup that holds the complete sum.
// flag: one int of LDS; call with all 256 threads of the 4 math waves (tid 0..255) after the k loop.
template <int TM, int TN>
__device__ __forceinline__ bool splitk_fold_acc(const GemmParams& p, f32x4 (&acc)[TM][TN], int tile_id, int zs, int* flag, int tid) {
    constexpr int NV = TM * TN;
    f32x4* __restrict__ slab = reinterpret_cast<f32x4*>(p.partial) + ((long)tile_id * p.splits) * (NV * 256) + tid;
    // the two counters of a tile are agent-scope words in BOTH forms (as the counters of round 2's XCD-local fold were): an L2-scope load in the wait loop below
    // may be served by the CU's L1 for ever -- the first version of this function hung there
    unsigned* w = reinterpret_cast<unsigned*>(p.tickets) + 2 * tile_id;   // [0] arrivals, [1] publications
    __syncthreads();                                                       // (every wave is done reading the LDS ring: `flag` lives in its first bytes)
    if (tid == 0) *flag = (int)__hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool is_last = *flag == p.splits - 1;
    if (!is_last) {
        f32x4* dst = slab + (long)zs * (NV * 256);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + (i * TN + j) * 256), "v"(acc[i][j]) : "memory");   // write-through: past the XCD's L2
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(w + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
    }
    if (tid == 0) {
        // the publishers HAVE arrived (they took their tickets before this workgroup did), so they are running and need nobody -- still, the wait is bounded
        // (20 ms of the 100 MHz wall clock, which keeps counting while a wave is descheduled: long enough for queue pre-emption / two processes time-slicing one
        // GPU): should a publication never come, the pass is marked invalid (the host-mapped flag osg_sync / osg_download check) instead of hanging the device
        const unsigned want = (unsigned)p.splits - 1;
        const unsigned long long t0 = wall_clock64();
        bool complete = true;
        while (__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
            if (wall_clock64() - t0 > 2000000ull) {
                if (p.xcd_err) __hip_atomic_store(p.xcd_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                complete = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        // both words back to zero for the next launch (nobody else touches them again: every other slice has published and left).  NOT after a timeout (advisor,
        // round 5): a late publisher would bump the zeroed word and every later launch of this tile would fold early -- the host zeroes ALL counters when it sees
        // the flag (osg_ctx.hip xcd_check), behind everything the stream holds
        if (complete) {
            __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    f32x4 sum[TM][TN];
    for (int s = 0; s < p.splits; s++) {
        if (s == zs) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) sum[i][j] = s == 0 ? acc[i][j] : sum[i][j] + acc[i][j];
        } else {
            const f32x4* src = slab + (long)s * (NV * 256);
            f32x4 part[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[i][j]) : "v"(src + (i * TN + j) * 256) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    asm volatile("" : "+v"(part[i][j]));
                    sum[i][j] = s == 0 ? part[i][j] : sum[i][j] + part[i][j];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = sum[i][j];
    return true;
}

int launch_splitk_reduce(osg_ctx* ctx, const GemmParams& p, int batch);   // osg_gemm.hip
// osg_gemm_wide.hip (round 6): tiles 4 .. 7 of the direct-to-LDS kernel -- 128 x 160, 128 x 80, 64 x 80, 64 x 160; -2 = no such instantiation
int launch_v2_wide(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst, bool conv, int spec = 0);   // spec: four loader waves (tile 4, 4-stage ring)
bool wide_tile_has(int tile, int nst, bool conv, bool ln1, bool ln2, bool geglu, bool rowstats);
// the statistics a StatSink asks for, from the stored output (rows ldc apart) -- for the launches whose epilogue does not serve sinks (osg_norm.hip)
int launch_colstats(osg_ctx* ctx, const f16* C, long ldc, int M, int N, int rows_per_image, const StatSink* sinks);
long long* kdbg_buffer(osg_ctx* ctx, long workgroups);   // osg_ctx.hip: NULL unless OSG_KDBG is set
inline int no_epi_prefetch() { static const int v = getenv("OSG_NO_EPI_PREFETCH") ? 1 : 0; return v; }
// OSG_SPLITK_FOLD (round 5; read per call): 0 = never; 1 (default) = splitk_fold_acc where it applies
inline int splitk_fold_mode() {
    const char* e = getenv("OSG_SPLITK_FOLD");
    return e ? atoi(e) : 1;
}
// a launch of n_tiles output tiles x p.splits k-slices whose tiles are bm x bn: can it finish with splitk_fold_acc?  Sets tickets / fold_acc and returns the
// slab bytes the launch needs (0: no)
inline size_t splitk_fold_route(osg_ctx* ctx, GemmParams& p, long n_tiles, int bm, int bn) {
    p.fold_acc = 0;
    const int mode = splitk_fold_mode();
    if (mode == 0 || p.splits < 2 || p.splits > 4 || !ctx->tickets || 2 * n_tiles + 16 > osg_ctx::kTickets / 2) return 0;
    if ((p.N & 3) != 0 || ((p.ldc | p.ldc2) & 3) != 0 || p.act == OSG_ACT_GEGLU || p.ln_c1 || p.rs_out) return 0;   // the compact fused epilogue only
    if (p.sink[0].table || p.sink[1].table) return 0;   // (GroupNorm statistics of a split launch come from the reduce launch)
    p.fold_acc = 1;
    p.tickets = ctx->tickets;                            // (the lower half of the counters; the upper one belongs to the GroupNorm clusters, osg_norm.hip)
    // (an XCD-local form -- slices of a tile on one XCD, slabs through its L2 -- needs workgroup b of EVERY launch on XCD b mod 8; that holds for an isolated launch
    // but not inside a pass: the dispatcher carries on round-robin from wherever the previous kernel stopped, and the per-workgroup XCC_ID check failed the first
    // captured pass, profiles/r05_fold_xcd_local_fails_in_the_pass.txt.  Removed.)
    p.xcd_err = ctx->xcd_err_dev;       // (the bounded wait of the last arriver reports through it)
    return (size_t)n_tiles * p.splits * bm * bn * sizeof(float);
}
}  // namespace osg_mm

// osg_conv3x3.hip: halo-reuse 3x3 / stride 1 / pad 1 convolution.  Returns -1 when the shape is not one it takes.
int osg_conv3x3_run(osg_ctx* ctx, osg_mm::GemmParams& p);
// the same in pieces, for the measured configuration choice (osg_tune.h): shape gate, ranked (BN, splits) candidates, one launch
int osg_conv3x3_prepare(osg_ctx* ctx, osg_mm::GemmParams& p);
std::vector<std::pair<double, std::pair<int, int>>> osg_conv3x3_rank(const osg_ctx* ctx, const osg_mm::GemmParams& p);
int osg_conv3x3_launch(osg_ctx* ctx, osg_mm::GemmParams p, int bn, int splits, int loader_waves = 4, int fold = 0);   // fold: a 2 .. 4-way split finished by splitk_fold_acc
int osg_conv3x3_supported(int N, int H, int W, int Cin, int Cout);
// osg_conv3x3_w8.hip: the WQ = 1 instantiations of the halo kernel (uint8 weight codes, 4 loader waves); p.W in {64, 32, 16, 8}, bn in {80, 128, 160}
int osg_conv3x3_w8_tile(osg_ctx* ctx, osg_mm::GemmParams& p, int bn);
namespace osg_mm {
// osg_gemm_w8.hip: the WQ = 1 instantiations of gemm2_kernel; -2 when the (tile, ring, form) asked for has none (tiles as kV2BM / kV2BN of osg_gemm.hip)
int launch_v2_w8(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst, bool conv);
bool w8_tile_has(int tile, int nst, bool conv);
}
