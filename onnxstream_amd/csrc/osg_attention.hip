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
#include <type_traits>
#include <utility>

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

// =====================================================================================================================================
// v2 (round 5) -- the UNet's attention shapes (no mask, head dim 40 / 64 / 80 / 160).  Same mapping and the same online softmax as attn_kernel,
// but the per-tile instruction count of a wave is roughly halved (the kernel is issue-bound: ~380 instructions per 64-key tile and wave around 28 MFMAs):
//   * K / V tiles arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction) into an NST-deep ring -- no staging registers, no ds_write pass,
//     ONE barrier per tile; rows past Tkv are zero-filled by the descriptor's bounds check (out-of-range offset);
//   * the LDS image of a tile is the row-major [64 keys][RSC 16-byte chunks] copy (lane-linear, as the DMA requires), rows PADDED to RSC = 2 (mod 4) chunks --
//     6 for D = 40, 10 for 64 / 80, 22 for 160 (the pad lanes request an out-of-range offset: zeros).  With that stride both fragment reads are free of bank
//     conflicts without a swizzle: ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27} ..., i.e. rows {0-3, 12-15} at chunk c with rows 4-11 at
//     chunk c + 1, whose 16-byte slots (row RSC + chunk) mod 16 are then all distinct; the transposing read's 32-lane groups cover 8 rows x 32 bytes at
//     distinct multiples of 32 bytes mod 256.  (Dense 80-byte rows: half of all LDS cycles were conflicts, profiles/r05_pmc_attention_v2.txt.)  V stays
//     row-major and is read with the transposing LDS read (ds_read_b64_tr_b16: a 16-lane group hands in 4 rows x 16 columns and each lane gets ONE column's
//     4 rows), which yields exactly the V^T fragment the P V step needs in the k order the probabilities already have -- the 8 ds_write_b16 per chunk of v1 are gone;
//   * the row sums come off the matrix pipe: an MFMA with an all-ones A operand adds up the (f16) probabilities of every query column -- 16 v_add_f32 per
//     query tile and key tile become 2 MFMAs, and the sum is the sum of exactly the values P V multiplies;
//   * XCD-local placement: the flat grid is remapped so that all query blocks of one (image, head) run on ONE XCD (its L2 then holds that head's K / V once;
//     with blockIdx.y = head every XCD pulled every head: 94.5 MB fetched per launch against 21 MB of operands, profiles/r04_pmc_tuned_plan.json).
template <class F, int... I>
__device__ __forceinline__ void attn2_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N>
__device__ __forceinline__ void attn2_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the transposing LDS read as inline asm: through the builtin, hipcc 7.2 puts an s_waitcnt vmcnt(0) in front of every such read (it cannot tell the read from the
// LDS-DMA writes in flight), which would serialise the ring.  The caller waits (counted lgkmcnt) before it uses the result.
typedef __fp16 attn2_hx4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
template <int OFF>
__device__ __forceinline__ attn2_hx4 attn2_tr_read(unsigned lds_addr) {
    attn2_hx4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
    return v;
}

template <int D, int QT, int NST>
__global__ __launch_bounds__(256) void attn2_kernel(AttnParams pk, int nqb) {
    const AttnParams p = pk;
    osg_pin_all(p.q, p.k, p.v, p.o, p.q_tok, p.q_head, p.q_batch, p.k_tok, p.k_head, p.k_batch, p.v_tok, p.v_head, p.v_batch, p.o_tok, p.o_head, p.o_batch, p.heads, p.Tq, p.Tkv,
                p.scale_log2e, p.mask, p.inv_scale, p.kv_div, nqb, (int)gridDim.x);
    constexpr int DCH = D / 8;                  // 16-byte chunks per K / V row
    constexpr int RSC = DCH + ((6 - DCH % 4) % 4);   // ... of the LDS image: the next count = 2 (mod 4), see the header
    constexpr int ROWB = RSC * 16;              // bytes per row of the LDS image
    constexpr int TILE_B = 64 * ROWB;           // one operand's tile: 64 keys (= RSC KiB)
    constexpr int STAGE_B = 2 * TILE_B;         // K tile, then V tile
    constexpr int K32 = D / 32, REM = D % 32;
    static_assert(REM == 0 || REM == 8 || REM == 16, "attn2: head dim = 32 a (+ 8 | 16)");
    static_assert(RSC % 4 == 2 && RSC >= DCH && (REM == 0 || RSC * 8 >= K32 * 32 + 16), "row pitch");
    constexpr bool TAIL = REM != 0;
    constexpr int DT = (D + 15) / 16;
    static_assert(DT * 16 <= RSC * 8, "the P V row blocks read inside the padded row");
    constexpr int NT = 4, NS = 2;               // 16-key score tiles / 32-deep P V steps per 64-key tile
    constexpr int NINST = 2 * RSC;              // 1-KiB DMA pieces per tile: K's RSC, then V's RSC
    constexpr int MAXP = NINST / 4;             // ... per wave (RSC is even: the same count for every wave)
    static_assert((NST == 3 || NST == 4) && MAXP * (NST - 2) <= 63, "three or four stages (the tile being read, the next one, one or two in flight: sync_issue's tail guard `kt + 2 >= ntiles` covers no deeper ring); vmcnt is a 6-bit counter");
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char asmem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef attn2_hx4 hx4;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, g = lane >> 4;
    // flat workgroup index -> (image, head, query block): a contiguous run of indices per XCD (workgroup b runs on XCD b % 8)
    int L;
    {
        const int total = gridDim.x, bid = blockIdx.x, x = bid & 7, i = bid >> 3, q = total >> 3, r = total & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int bh = L / nqb, qb = L - bh * nqb;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qb * (64 * QT) + wave * (16 * QT);

    const f16* __restrict__ Q = p.q + b * p.q_batch + h * p.q_head;
    const f16* __restrict__ K = p.k + b * p.k_batch + (h / p.kv_div) * p.k_head;
    const f16* __restrict__ V = p.v + b * p.v_batch + (h / p.kv_div) * p.v_head;
    f16* __restrict__ O = p.o + b * p.o_batch + h * p.o_head;

    // buffer descriptors over this head's K / V rows; validity of a row is decided per lane (out-of-range offset => the DMA writes zeros)
    const long kext = ((long)(p.Tkv - 1) * p.k_tok + D) * 2, vext = ((long)(p.Tkv - 1) * p.v_tok + D) * 2;
    __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)(kext < 0x7ffffff0L ? kext : 0x7ffffff0L), 0x00020000);
    __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)(vext < 0x7ffffff0L ? vext : 0x7ffffff0L), 0x00020000);

    // this wave's DMA pieces: piece i = wave + 4 j covers bytes [i KiB, (i + 1) KiB) of the stage; lane l fills slot (i % RSC) * 64 + l of that operand's tile
    int p_off[MAXP], p_row[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; j++) {
        const int i = wave + 4 * j;
        const bool isv = i >= RSC;
        const int c = (isv ? i - RSC : i) * 64 + lane;
        const int row = c / RSC, pp = c - row * RSC;
        p_row[j] = pp < DCH ? row : 64;             // (a pad chunk is never in range)
        p_off[j] = (int)(((long)row * (isv ? p.v_tok : p.k_tok) + pp * 8) * 2);
    }
    const int ntiles = (p.Tkv + 63) >> 6;
    auto issue_tile = [&](int kt) __attribute__((always_inline)) {
        char* st = asmem + (kt % NST) * STAGE_B;
        const int kv0 = kt << 6, left = min(p.Tkv - kv0, 64);
        const int sk = (int)((long)kv0 * p.k_tok * 2), sv = (int)((long)kv0 * p.v_tok * 2);
#pragma unroll
        for (int j = 0; j < MAXP; j++) {
            const int i = wave + 4 * j;
            const unsigned off = p_row[j] < left ? (unsigned)p_off[j] : OOB;
            if (i < RSC) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr)(st + i * 1024), 16, off, sk, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr)(st + i * 1024), 16, off, sv, 0, 0);
        }
    };

    // Q fragments (operand B: lane holds Q[q][ks*32 + g*8 .. +7]; a partial last step: zero past D)
    f16x8 qf[QT][K32 ? K32 : 1];
    f16x8 qtail[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int q = q0 + qt * 16 + lq;
        const f16* qrow = Q + (long)min(q, p.Tq - 1) * p.q_tok;
#pragma unroll
        for (int ks = 0; ks < K32; ks++) qf[qt][ks] = *reinterpret_cast<const f16x8*>(qrow + ks * 32 + g * 8);
        if constexpr (TAIL) {   // the last, partial 32-deep step: Q[q][K32*32 + g*8 .. +7], zero past D
            const int d = K32 * 32 + g * 8;
            f16x8 v = *reinterpret_cast<const f16x8*>(qrow + min(d, D - 8));
            if (d >= D) v = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            qtail[qt] = v;
        }
    }
    // (rows past Tq compute on a copy of the last row and are not stored)
#pragma unroll
    for (int s2 = 0; s2 < NST - 1; s2++)
        if (s2 < ntiles) issue_tile(s2);

    f32x4 oacc[QT][DT], lacc[QT];
    float m_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        m_run[qt] = -INFINITY;
        lacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; dt++) oacc[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f16x8 ones = {(f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1};

    // fragment addressing inside a stage (constant over the tiles)
    int kofs[K32 + 1];
#pragma unroll
    for (int ks = 0; ks < K32 + 1; ks++) kofs[ks] = lq * ROWB + (ks * 4 + g) * 16;
    const int vofs = TILE_B + (g * 4 + (lq >> 2)) * ROWB + (lq & 3) * 8;
    const float c = p.scale_log2e;

    // ---- the sweep, software-pipelined over the key tiles: while the VALU runs the softmax of tile kt, the matrix pipe already works on S^T of tile kt + 1
    // (two score register sets).  The loop is unrolled so that the stage of a tile and the score set are compile-time: LDS addresses are one VGPR + immediates.
    auto qk = [&](auto stc, f32x4 (&s)[QT][NT]) __attribute__((always_inline)) {
        constexpr int stg = decltype(stc)::value;
        const char* St = asmem + stg * STAGE_B;
#pragma unroll
        for (int qt = 0; qt < QT; qt++)
#pragma unroll
            for (int t = 0; t < NT; t++) s[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // A head dim that is not a multiple of 32 ends with a partial step whose Q fragment is zero past D; the K fragment of that step reads on into the row's
        // zero pad and, past the row pitch, into the next row (finite data times zero).  It is a 32-deep MFMA like the others: v_mfma_f32_16x16x16_f16 occupies
        // the matrix pipe just as long (measured), its 64-bit fragment reads were merged into ds_read2_b64 -- 4-way bank conflicts, ALL the conflict cycles of the
        // kernel -- and hipcc 7.2 pads no wait states between a 32-deep MFMA and a 16-deep one that reads its accumulator (wrong scores when they were adjacent).
#pragma unroll
        for (int ks = 0; ks < K32 + (TAIL ? 1 : 0); ks++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(St + t * 16 * ROWB + kofs[ks]);
#pragma unroll
                for (int qt = 0; qt < QT; qt++) s[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, ks < K32 ? qf[qt][ks < K32 ? ks : 0] : qtail[qt], s[qt][t], 0, 0, 0);
            }
    };
    // online softmax (log2 domain), as attn_kernel: raw running maximum, the scale rides in the fma that feeds v_exp_f32
    // (fullc: every key of the tile exists -- all tiles but a ragged last one.  The rescale of the accumulators stays behind a ballot: unconditional it is 25
    // VALU instructions per tile in a kernel bound by VALU issue -- measured 103.3 vs 101.4 us, although the branch-free step let the scheduler lay the
    // softmax between the next tile's MFMAs.)
    auto softmax = [&](auto fullc, f32x4 (&s)[QT][NT], int kv0, f16x8 (&pf)[QT][NS]) __attribute__((always_inline)) {
        constexpr bool full = decltype(fullc)::value;
#pragma unroll
        for (int qt = 0; qt < QT; qt++) {
            float mx = -INFINITY;
            if constexpr (full) {
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
            const float m_new = fmaxf(m_run[qt], mx);
            const float mc = -m_new * c;
            if (__builtin_amdgcn_ballot_w64(m_new != m_run[qt]) != 0) {   // (exactly 1 everywhere otherwise; the maxima settle after a few tiles)
                const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * c);
#pragma unroll
                for (int dt = 0; dt < DT; dt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) oacc[qt][dt][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 4; r++) lacc[qt][r] *= alpha;
            }
            m_run[qt] = m_new;
#pragma unroll
            for (int st = 0; st < NS; st++) {
                f16x8 pv;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pv[r] = (f16)__builtin_amdgcn_exp2f(fmaf(s[qt][2 * st][r], c, mc));
                    pv[4 + r] = (f16)__builtin_amdgcn_exp2f(fmaf(s[qt][2 * st + 1][r], c, mc));
                }
                pf[qt][st] = pv;
            }
        }
    };
    // O^T += V^T P, l += 1^T P.  V^T fragments: k slots g*8 + 0..3 <-> keys st*32 + g*4 + 0..3, slots g*8 + 4..7 <-> keys st*32 + 16 + g*4 + 0..3 (the order of
    // pf); the reads of row block dt + 1 are in flight while block dt's MFMAs issue (LDS returns in order: a counted wait)
    auto pvstep = [&](auto stc, f16x8 (&pf)[QT][NS]) __attribute__((always_inline)) {
        constexpr int stg = decltype(stc)::value;
#pragma unroll
        for (int st = 0; st < NS; st++)
#pragma unroll
            for (int qt = 0; qt < QT; qt++) lacc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[qt][st], lacc[qt], 0, 0, 0);
        // (the stage's base rides in the 16-bit immediate where it fits, in the address register otherwise)
        constexpr int SB = stg * STAGE_B, IMM = (SB + 48 * ROWB + DT * 32 < 65536) ? SB : 0;
        const unsigned va = (unsigned)(size_t)(lds_ptr)(asmem + vofs + (SB - IMM));
        hx4 vr[2][NS][2];
        auto vread = [&](auto dtc) __attribute__((always_inline)) {
            constexpr int dt = decltype(dtc)::value;
            vr[dt & 1][0][0] = attn2_tr_read<IMM + 0 * ROWB + dt * 32>(va);
            vr[dt & 1][0][1] = attn2_tr_read<IMM + 16 * ROWB + dt * 32>(va);
            vr[dt & 1][1][0] = attn2_tr_read<IMM + 32 * ROWB + dt * 32>(va);
            vr[dt & 1][1][1] = attn2_tr_read<IMM + 48 * ROWB + dt * 32>(va);
        };
        auto vstep = [&](auto dtc) __attribute__((always_inline)) {
            constexpr int dt = decltype(dtc)::value;
            hx4 &a0 = vr[dt & 1][0][0], &a1 = vr[dt & 1][0][1], &a2 = vr[dt & 1][1][0], &a3 = vr[dt & 1][1][1];
            if constexpr (dt + 1 < DT) {
                vread(std::integral_constant<int, dt + 1>{});
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"memory");
            } else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)::"memory");
            // (two 16-deep MFMAs fed by one transposing read each would save the v_mov_b64 that glues two 64-bit results into the 128-bit operand below -- 18 VALU
            // instructions per tile -- but measured 109 vs 98 us: a v_mfma_f32_16x16x16_f16 occupies the matrix pipe as long as the 32-deep one)
#pragma unroll
            for (int st = 0; st < NS; st++) {
                const hx4 lo = vr[dt & 1][st][0], hi = vr[dt & 1][st][1];
                const f16x8 vf = {(f16)lo[0], (f16)lo[1], (f16)lo[2], (f16)lo[3], (f16)hi[0], (f16)hi[1], (f16)hi[2], (f16)hi[3]};
#pragma unroll
                for (int qt = 0; qt < QT; qt++)
                    oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][st], oacc[qt][dt], 0, 0, 0);
            }
        };
        vread(std::integral_constant<int, 0>{});
        attn2_static_for(vstep, std::make_integer_sequence<int, DT>{});
    };
    // before tile kt + 1 is read: my pieces of it have landed (the NST - 3 tiles requested after it may stay in flight), then everyone's have -- and
    // everyone is done with tile kt - 1, whose stage takes tile kt + NST - 1
    auto sync_issue = [&](int kt) __attribute__((always_inline)) {
        if (NST == 3 || kt + 2 >= ntiles) attn2_wait_vm<0>();
        else attn2_wait_vm<MAXP * (NST - 3)>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < ntiles) issue_tile(kt + NST - 1);
    };
    f32x4 sA[QT][NT], sB[QT][NT];
    sync_issue(-1);                                   // tile 0 (no issue: tiles 0 .. NST - 2 are already requested)
    qk(std::integral_constant<int, 0>{}, sA);
    constexpr int U = (NST % 2) ? 2 * NST : NST;      // unroll: stage and score set repeat with this period
    // a step of the sweep: tiles 0 .. ntiles - 2 (full tiles with a successor), then the last one (maybe ragged, nothing to prefetch)
    auto step = [&](auto ic, int kt) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        f32x4 (&sc)[QT][NT] = (i & 1) ? sB : sA;
        f32x4 (&sn)[QT][NT] = (i & 1) ? sA : sB;
        f16x8 pf[QT][NS];
        sync_issue(kt);
        qk(std::integral_constant<int, (i + 1) % NST>{}, sn);
        softmax(std::true_type{}, sc, kt << 6, pf);
        pvstep(std::integral_constant<int, i % NST>{}, pf);
    };
    auto last = [&](auto ic, int kt) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        f32x4 (&sc)[QT][NT] = (i & 1) ? sB : sA;
        f16x8 pf[QT][NS];
        if ((p.Tkv & 63) == 0) softmax(std::true_type{}, sc, kt << 6, pf);
        else softmax(std::false_type{}, sc, kt << 6, pf);
        pvstep(std::integral_constant<int, i % NST>{}, pf);
    };
    // main loop: U unconditional steps per trip (guards inside the unrolled body make the register allocator shuffle the score sets at every join: 50 v_mov_b64
    // per tile were measured in such a build); what is left -- fewer than U steps and the last tile -- runs once, through guarded copies of the same code
    int kt = 0;
    for (; kt + U < ntiles; kt += U)
        attn2_static_for([&](auto ic) __attribute__((always_inline)) { step(ic, kt + decltype(ic)::value); }, std::make_integer_sequence<int, U>{});
    bool done = false;
    attn2_static_for([&](auto ic) __attribute__((always_inline)) {
        const int k2 = kt + decltype(ic)::value;
        if (!done) {
            if (k2 + 1 < ntiles) step(ic, k2);
            else { last(ic, k2); done = true; }
        }
    }, std::make_integer_sequence<int, U>{});

    // ---- epilogue: O[q][d..d+3] = O^T / l -----------------------------------------------------------------
#pragma unroll
    for (int qt = 0; qt < QT; qt++) {
        const int q = q0 + qt * 16 + lq;
        if (q >= p.Tq) continue;
        const float inv = 1.0f / lacc[qt][0];
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

template <int D, int NST>
int launch_attn2(osg_ctx* ctx, const AttnParams& p, int batch) {
    static const int force_qt = getenv("OSG_ATTN_QT") ? atoi(getenv("OSG_ATTN_QT")) : 0;
    constexpr int RSC = D / 8 + ((6 - (D / 8) % 4) % 4);
    constexpr size_t smem = (size_t)NST * 2 * 64 * RSC * 16;
    static_assert(smem <= 160 * 1024, "LDS budget");
    const long blocks128 = (long)((p.Tq + 127) / 128) * batch * p.heads;
    const bool qt2 = p.Tq >= 1024 && force_qt != 1 && (blocks128 >= 2L * ctx->num_cu || force_qt == 2);
    auto k1 = attn2_kernel<D, 1, NST>;
    auto k2 = attn2_kernel<D, 2, NST>;
    static unsigned long long attr_mask = 0;   // (per device: hipFuncSetAttribute is, and a process may hold several)
    if (osg_first_on_device(attr_mask)) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    const int nqb = qt2 ? (p.Tq + 127) / 128 : (p.Tq + 63) / 64;
    dim3 grid((unsigned)((long)nqb * batch * p.heads));
    if (qt2) hipLaunchKernelGGL(k2, grid, dim3(256), smem, ctx->compute, p, nqb);
    else hipLaunchKernelGGL(k1, grid, dim3(256), smem, ctx->compute, p, nqb);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
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
    // the UNet's shapes: v2 (OSG_ATTN_V1=1: the round-2 kernel, for A/B).  K / V rows must be 16-byte aligned (checked by the callers); the DMA offsets are 32-bit
    static const int v1 = getenv("OSG_ATTN_V1") ? atoi(getenv("OSG_ATTN_V1")) : 0;
    if (!v1 && !p.mask && (long)p.Tkv * max(p.k_tok, p.v_tok) * 2 < 0x7fffffffL) {
        if (D == 40) return launch_attn2<40, 4>(ctx, p, batch);
        if (D == 64) return launch_attn2<64, 3>(ctx, p, batch);
        if (D == 80) return launch_attn2<80, 3>(ctx, p, batch);
        if (D == 160) return launch_attn2<160, 3>(ctx, p, batch);
    }
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
