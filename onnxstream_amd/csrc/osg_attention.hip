// libosgpu: fused attention on the matrix cores (flash-style, online softmax), f16 operands / f32 accumulate.
//
// Replaces the reference's AttentionFusedOps pseudo-op (onnxstream.cpp:6696-6929), which runs, per head and per
// Q row-chunk ("sliced attention"), 4 XNNPACK operators and round-trips the score slab through memory:
//     S = Q K^T ; S *= scale ; P = softmax_rows(S) ; O = P V.
// Here the [Tq x Tkv] scores never leave the chip, so slicing (m_attention_fused_ops_parts) is unnecessary and the
// result is independent of it.
//
// Mapping (one workgroup = 4 waves = 64*QT query rows of one head; KV swept in tiles of 64):
//   S^T[kv][q] = mfma_16x16x32(A = K rows from LDS, B = Q fragment held in registers)      (contraction over d)
//   lane (q = lane&15, g = lane>>4) then owns scores kv = t*16 + g*4 + r: the row max / sum need only two
//   cross-lane steps (xor 16, 32); P is converted to f16 IN PLACE as the B operand of
//   O^T[d][q] += mfma_16x16x32(A = V^T fragment from LDS, B = P)                           (contraction over kv)
//   -- the kv order inside a 32-wide contraction step is permuted identically for P and V^T, so P never moves
//   between lanes.  The O^T layout gives each lane 4 consecutive d of one query row: 8-byte stores.
// Head dims are zero-padded in LDS/registers to a multiple of 32 (QK^T) / 16 (PV): 40->64/48, 80->96/80, 160->160.
#include "osg_common.h"
#include <cstdlib>

namespace {

struct AttnParams {
    const f16 *q, *k, *v;
    f16* o;
    long q_tok, q_head, q_batch;
    long k_tok, k_head, k_batch;
    long v_tok, v_head, v_batch;
    long o_tok, o_head, o_batch;
    int heads, Tq, Tkv, D;
    float scale_log2e;
    // ScaledDotProductAttention (osg_sdpa): additive mask [Tq][Tkv] shared by every batch and head (null = none), applied to the raw scores
    // as s + mask / scale (the kernel tracks raw maxima and scales inside the exponent); grouped-query attention: query head h reads
    // key/value head h / kv_div
    const f16* mask;
    float inv_scale;
    int kv_div;
};


// all-reduce over the 4 lanes {l, l^16, l^32, l^48} that share a query row, on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap: the odd
// 16-lane rows of one operand trade places with the even rows of the other, resp. the upper half with the lower half) -- the ds_bpermute a
// __shfl_xor compiles to is an LDS round trip of ~100 cycles, and the softmax chains four of them per query tile behind each other
// (inline asm: the __builtin_amdgcn_permlane*_swap pair comes back from hipcc 7.2 with both results folded into one register; s_nop 1 covers
// the VALU-write -> permlane-read hazard the compiler cannot see inside the asm)
__device__ __forceinline__ void lane_swap16(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap32(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float row4_max(float v) {
    float a, b;
    lane_swap16(v, a, b);
    lane_swap32(fmaxf(a, b), a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float row4_sum(float v) {
    float a, b;
    lane_swap16(v, a, b);
    lane_swap32(a + b, a, b);
    return a + b;
}

// BKV: keys per tile (64, or 128 where registers and LDS allow: half the barriers, running-max updates and accumulator rescales per key)
template <int DP, int DT, int QT, int BKV>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
    constexpr int NT = BKV / 16;    // 16-key score tiles per KV tile
    constexpr int NS = BKV / 32;    // 32-deep contraction steps of P V
    constexpr int RPL = BKV / 64;   // KV rows staged per lane
    constexpr int KLD = DP + 8;     // Ks row stride (halves): 16-byte aligned rows, conflict-free b128 reads
    constexpr int VLD = BKV + 8;    // Vt row stride (halves): 8-byte aligned b64 reads
    constexpr int DV = DT * 16;
    constexpr int KS = DP / 32;
    __shared__ __attribute__((aligned(16))) f16 Ks[BKV * KLD];
    __shared__ __attribute__((aligned(16))) f16 Vt[DV * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const int D = p.D;

    const f16* __restrict__ Q = p.q + b * p.q_batch + h * p.q_head;
    const f16* __restrict__ K = p.k + b * p.k_batch + (h / p.kv_div) * p.k_head;
    const f16* __restrict__ V = p.v + b * p.v_batch + (h / p.kv_div) * p.v_head;
    f16* __restrict__ O = p.o + b * p.o_batch + h * p.o_head;

    // zero the LDS padding once (pad columns of Ks, pad rows of Vt) so no NaN bit patterns enter the MFMAs
    for (int i = tid; i < BKV * KLD; i += 256) Ks[i] = (f16)0;
    for (int i = tid; i < DV * VLD; i += 256) Vt[i] = (f16)0;

    // Q fragments (operand B: lane holds Q[q][ks*32 + g*8 .. +7])
    f16x8 qf[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int q = q0 + qt * 16 + lq;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int d = ks * 32 + g * 8;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (q < p.Tq && d < D) v = *reinterpret_cast<const f16x8*>(Q + (long)q * p.q_tok + d);
            qf[qt][ks] = v;
        }
    }

    f32x4 oacc[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int dchunks = D / 8;
    // K/V staging: lane = kv row of the tile, the 4 waves stride over the 16-byte d-chunks.  The NEXT tile's global loads are issued
    // into registers BEFORE this tile's MFMAs and written to LDS after them, so their latency hides behind the math.
    constexpr int NCH = (DP / 8 + 3) / 4;            // d-chunks per wave (upper bound)
    f16x8 kreg[RPL][NCH], vreg[RPL][NCH];
    // loads are UNCONDITIONAL (row / chunk clamped into range) and the out-of-range zeroing happens at store time: a predicated load
    // makes hipcc wait for it right after the issue (it needs the value for the select), which serialises the prefetch
    // Rows past Tkv are NOT zeroed either: they hold copies of the last valid row (finite), their scores are masked to -inf on the partial tile,
    // so their probabilities are exactly 0 and the copies contribute exactly 0 to P V.
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int kv = min(kv0 + j * 64 + lane, p.Tkv - 1);
            const long krow = (long)kv * p.k_tok, vrow = (long)kv * p.v_tok;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int dc = min(wave + c * 4, dchunks - 1);
                kreg[j][c] = *reinterpret_cast<const f16x8*>(K + krow + dc * 8);
                vreg[j][c] = *reinterpret_cast<const f16x8*>(V + vrow + dc * 8);
            }
        }
    };
    // V^T columns are stored PERMUTED inside each 32-wide block -- kv = t*16 + g*4 + r  ->  st*32 + g*8 + (t&1)*4 + r -- which is the
    // order a lane's 8 probabilities come out of the S^T tiles, so the PV operand is ONE ds_read_b128 (no register shuffling)
    const int vcol = ((lane >> 5) << 5) + (((lane >> 2) & 3) << 3) + (((lane >> 4) & 1) << 2) + (lane & 3);
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < RPL; j++)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int dc = wave + c * 4;
                if (dc < dchunks) {
                    const f16x8 kv8 = kreg[j][c], vv8 = vreg[j][c];
                    *reinterpret_cast<f16x8*>(&Ks[(j * 64 + lane) * KLD + dc * 8]) = kv8;
#pragma unroll
                    for (int e = 0; e < 8; e++) Vt[(dc * 8 + e) * VLD + j * 64 + vcol] = vv8[e];
                }
            }
    };
    load_tile(0);
    for (int kv0 = 0; kv0 < p.Tkv; kv0 += BKV) {
        __syncthreads();  // previous tile fully consumed (also orders the initial zero fill)
        store_tile();
        __syncthreads();
        if (kv0 + BKV < p.Tkv) load_tile(kv0 + BKV);   // in flight during the MFMAs / softmax below

        // ---- S^T = K Q^T -------------------------------------------------------------------------------
        f32x4 s[QT][NT];
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
#pragma unroll
            for (int t = 0; t < NT; t++) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                f16x8 kf = *reinterpret_cast<const f16x8*>(&Ks[(t * 16 + lq) * KLD + ks * 32 + g * 8]);
#pragma unroll
                for (int qt = 0; qt < QT; qt++) s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], s[qt][t], 0, 0, 0);
            }
        }

        // ---- online softmax (log2 domain).  The kernel is VALU-bound here (28 MFMAs vs ~300 vector ops per tile and wave), so the
        // per-score work is kept to max / fma / v_exp_f32 / add: the scale rides in the fma (scores stay raw, max is tracked raw),
        // the kv-bound mask only exists on the last, partial tile, and exp2 is the bare hardware op (arguments are <= 0).
        if (p.mask) {   // (not on the SD hot path: scalar f16 loads, clamped into range -- out-of-range scores are masked below anyway)
#pragma unroll
            for (int qt = 0; qt < QT; qt++) {
                const f16* mrow = p.mask + (long)min(q0 + qt * 16 + lq, p.Tq - 1) * p.Tkv;
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        s[qt][t][r] = fmaf((float)mrow[min(kv0 + t * 16 + g * 4 + r, p.Tkv - 1)], p.inv_scale, s[qt][t][r]);
            }
        }
        const bool full = kv0 + BKV <= p.Tkv;
        const float c = p.scale_log2e;
        f16x8 pf[QT][NS];
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            float mx = -INFINITY;
            if (full) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) mx = fmaxf(mx, s[qt][t][r]);
            } else {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int kv = kv0 + t * 16 + g * 4 + r;
                        const float x = kv < p.Tkv ? s[qt][t][r] : -INFINITY;
                        s[qt][t][r] = x;
                        mx = fmaxf(mx, x);
                    }
            }
            mx = row4_max(mx);
            const float m_new = fmaxf(m_run[qt], mx);             // raw (unscaled) running max; scale > 0
            const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * c);
            const float mc = -m_new * c;
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[qt][t][r], c, mc));
                    s[qt][t][r] = e;
                    sum += e;
                }
            sum = row4_sum(sum);
            l_run[qt] = l_run[qt] * alpha + sum;
            // the running maximum settles after a few tiles: when no row of this wave moved, alpha is exactly 1 everywhere and the rescale is skipped
            if (__builtin_amdgcn_ballot_w64(m_new != m_run[qt]) != 0) {
#pragma unroll
                for (int dt = 0; dt < DT; dt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) oacc[qt][dt][r] *= alpha;
            }
            m_run[qt] = m_new;
#pragma unroll
            for (int st = 0; st < NS; st++) {
                f16x8 pv;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pv[r] = (f16)s[qt][2 * st][r];
                    pv[4 + r] = (f16)s[qt][2 * st + 1][r];
                }
                pf[qt][st] = pv;
            }
        }

        // ---- O^T += V^T P ------------------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
#pragma unroll
            for (int st = 0; st < NS; st++) {
                const f16x8 vf = *reinterpret_cast<const f16x8*>(&Vt[(dt * 16 + lq) * VLD + st * 32 + g * 8]);
#pragma unroll
                for (int qt = 0; qt < QT; qt++)
                    oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][st], oacc[qt][dt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: O[q][d..d+3] = O^T / l -----------------------------------------------------------------
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int q = q0 + qt * 16 + lq;
        if (q >= p.Tq) continue;
        const float inv = 1.0f / l_run[qt];
#pragma unroll
        for (int dt = 0; dt < DT; dt++) {
            const int d = dt * 16 + g * 4;
            if (d >= D) continue;
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = (f16)(oacc[qt][dt][r] * inv);
            *reinterpret_cast<f16x4*>(O + (long)q * p.o_tok + d) = o;
        }
    }
}

template <int DP, int DT>
int launch_attn(osg_ctx* ctx, const AttnParams& p, int batch) {
    static const int force_qt = getenv("OSG_ATTN_QT") ? atoi(getenv("OSG_ATTN_QT")) : 0;
    // 128 query rows per workgroup halve the K/V staging per row, but only pay when the grid still has >= 2 workgroups per CU
    const long blocks128 = (long)((p.Tq + 127) / 128) * batch * p.heads;
    static const int force_bkv = getenv("OSG_ATTN_BKV") ? atoi(getenv("OSG_ATTN_BKV")) : 0;
    // 128-key tiles (OSG_ATTN_BKV=128, head dims <= 64 only): measured SLOWER -- 171 vs 125 us at 4096x4096x40: the score tile and the staging
    // registers double (255 VGPRs), and what a tile saves in barriers it loses in exposed latency.  Kept as an experiment switch.
    constexpr bool kWide = DP <= 64;
    const bool wide = kWide && force_bkv == 128;
    if (p.Tq >= 1024 && force_qt != 1 && (blocks128 >= 2L * ctx->num_cu || force_qt == 2)) {
        dim3 grid((p.Tq + 127) / 128, batch * p.heads);
        if constexpr (kWide) {
            if (wide) hipLaunchKernelGGL((attn_kernel<DP, DT, 2, 128>), grid, dim3(256), 0, ctx->compute, p);
            else hipLaunchKernelGGL((attn_kernel<DP, DT, 2, 64>), grid, dim3(256), 0, ctx->compute, p);
        } else hipLaunchKernelGGL((attn_kernel<DP, DT, 2, 64>), grid, dim3(256), 0, ctx->compute, p);
    } else {
        dim3 grid((p.Tq + 63) / 64, batch * p.heads);
        if constexpr (kWide) {
            if (wide) hipLaunchKernelGGL((attn_kernel<DP, DT, 1, 128>), grid, dim3(256), 0, ctx->compute, p);
            else hipLaunchKernelGGL((attn_kernel<DP, DT, 1, 64>), grid, dim3(256), 0, ctx->compute, p);
        } else hipLaunchKernelGGL((attn_kernel<DP, DT, 1, 64>), grid, dim3(256), 0, ctx->compute, p);
    }
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int dispatch_attn(osg_ctx* ctx, const AttnParams& p, int batch) {
    const int D = p.D;
    if (D <= 32) return launch_attn<32, 2>(ctx, p, batch);
    if (D <= 48) return launch_attn<64, 3>(ctx, p, batch);
    if (D <= 64) return launch_attn<64, 4>(ctx, p, batch);
    if (D <= 80) return launch_attn<96, 5>(ctx, p, batch);
    if (D <= 96) return launch_attn<96, 6>(ctx, p, batch);
    if (D <= 128) return launch_attn<128, 8>(ctx, p, batch);
    if (D <= 160) return launch_attn<160, 10>(ctx, p, batch);
    OSG_FAIL(ctx, "osg_attention: head dim > 160 not implemented");
}

}  // namespace

extern "C" {

int osg_attention_strided(osg_ctx* ctx, osg_dtype dtype, const void* q, long q_tok, long q_head, long q_batch, const void* k, long k_tok,
                          long k_head, long k_batch, const void* v, long v_tok, long v_head, long v_batch, void* o, long o_tok,
                          long o_head, long o_batch, int batch, int heads, int Tq, int Tkv, int D, float scale) {
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_attention: only f16 arithmetic is implemented on the device");
    if (batch <= 0 || heads <= 0 || Tq <= 0 || Tkv <= 0 || D <= 0) OSG_FAIL(ctx, "osg_attention: invalid shape(s) of q, k and/or v");
    if (!(scale > 0.f)) OSG_FAIL(ctx, "osg_attention: the fused kernel tracks the row maximum of the raw scores and needs scale > 0");
    if (D % 8) OSG_FAIL(ctx, "osg_attention: head dim must be a multiple of 8");
    if ((q_tok | q_head | q_batch | k_tok | k_head | k_batch | v_tok | v_head | v_batch) % 8 || (o_tok | o_head | o_batch) % 4)
        OSG_FAIL(ctx, "osg_attention: strides must keep 16-byte (q,k,v) / 8-byte (o) alignment");
    AttnParams p{(const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, q_tok, q_head, q_batch, k_tok, k_head, k_batch,
                 v_tok, v_head, v_batch, o_tok, o_head, o_batch, heads, Tq, Tkv, D, scale * 1.4426950408889634f, nullptr, 0.f, 1};
    return dispatch_attn(ctx, p, batch);
}

// ScaledDotProductAttention (reference op src/onnxstream.cpp:7767-7882 -> XnnPack::scaled_dot_product_attention :2054-2150): dense
// q [B][Hq][Tq][D], k / v [B][Hkv][Tkv][D], optional additive mask [Tq][Tkv], out [B][Hq][Tq][D]; Hq % Hkv == 0.
int osg_sdpa(osg_ctx* ctx, osg_dtype dtype, const void* q, const void* k, const void* v, const void* mask, void* o, int batch, int q_heads,
             int kv_heads, int Tq, int Tkv, int D, float scale) {
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_sdpa: only f16 arithmetic is implemented on the device");
    if (batch <= 0 || q_heads <= 0 || kv_heads <= 0 || Tq <= 0 || Tkv <= 0 || D <= 0) OSG_FAIL(ctx, "ScaledDotProductAttention: invalid shape of query, key or value.");
    if (q_heads % kv_heads) OSG_FAIL(ctx, "ScaledDotProductAttention: query heads must be a multiple of key/value heads.");
    if (!(scale > 0.f)) OSG_FAIL(ctx, "osg_sdpa: the fused kernel tracks the row maximum of the raw scores and needs scale > 0");
    if (D % 8) OSG_FAIL(ctx, "osg_sdpa: head dim must be a multiple of 8");
    const long qh = (long)Tq * D, kh = (long)Tkv * D;
    AttnParams p{(const f16*)q, (const f16*)k, (const f16*)v, (f16*)o, D, qh, qh * q_heads, D, kh, kh * kv_heads, D, kh, kh * kv_heads,
                 D, qh, qh * q_heads, q_heads, Tq, Tkv, D, scale * 1.4426950408889634f, (const f16*)mask, 1.0f / scale, q_heads / kv_heads};
    return dispatch_attn(ctx, p, batch);
}

int osg_attention(osg_ctx* ctx, osg_dtype dtype, const void* q, const void* k, const void* v, void* o, int heads, int Tq, int Tkv, int D,
                  float scale, int k_is_dt) {
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_attention: only f16 arithmetic is implemented on the device");
    const void* kk = k;
    if (k_is_dt) {
        // the reference hands K over already transposed ([heads, D, Tkv], onnxstream.cpp:6792): undo it into scratch
        size_t bytes = (size_t)heads * D * Tkv * sizeof(f16);
        if (osg_ensure_workspace2(ctx, bytes)) return 1;
        long shape[3] = {heads, D, Tkv};
        int perm[3] = {0, 2, 1};
        if (osg_transpose(ctx, 2, k, ctx->ws2, 3, shape, perm)) return 1;
        kk = ctx->ws2;
    }
    return osg_attention_strided(ctx, dtype, q, D, (long)Tq * D, 0, kk, D, (long)Tkv * D, 0, v, D, (long)Tkv * D, 0, o, D, (long)Tq * D, 0,
                                 1, heads, Tq, Tkv, D, scale);
}

}  // extern "C"
