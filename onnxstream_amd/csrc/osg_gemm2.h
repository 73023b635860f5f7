// libosgpu: the direct-to-LDS pipelined contraction kernel (gemm2_kernel) and its launcher, shared by the translation units that instantiate it
// (osg_gemm.hip: the 128x128 / 128x64 / 64x64 / 64x128 tiles; osg_gemm_wide.hip, round 6: the 160- and 80-column tiles).
#pragma once
#include "osg_gemm_common.h"
#include "osg_tune.h"
#include <type_traits>

#ifndef OSG_GEMM_PIN
#define OSG_GEMM_PIN 1      // the kernel-argument fields of the prologue pulled in one batch at entry (GemmParams); 0 = as before (A/B)
#endif

namespace {
using namespace osg_mm;

// =====================================================================================================================
// v2: direct-to-LDS pipelined kernel (the hot one).  Requirements: K % 64 == 0 (conv: Cin % 64 == 0), 16-byte aligned rows.
//   * both operands stream HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write pass); the
//     buffer descriptor's bounds check zero-fills M/N tails AND the convolution halo (out-of-image taps get an
//     out-of-range offset), so the inner loop has no predication at all.
//   * BK = 64: one tile row = 128 B = 8 x 16-B chunks; chunk c of row r lives at slot c ^ (r & 7) (XOR swizzle applied
//     on the per-lane SOURCE address, LDS image stays lane-linear as the DMA requires) => conflict-free ds_read_b128.
//   * NST-deep LDS ring, counted `s_waitcnt vmcnt(N)` (never 0 in the loop), ONE raw s_barrier per k-tile; one workgroup
//     per CU owns up to 128 KiB of the 160 KiB LDS: with grids of ~1 tile per CU the latency hiding has to come from the
//     depth of the ring, not from co-resident blocks.
//   * XCD-aware tile walk: the 1-D grid is remapped so each of the 8 XCDs (private L2) gets a CONTIGUOUS run of tiles,
//     ordered so that the operand with more unique bytes is split across XCDs and read from HBM once.
// SPEC = 1: 512 threads -- waves 4..7 only issue the DMA loads, waves 0..3 only do ds_read + MFMA + epilogue (see osg_conv3x3.hip)
// LN = 1: LayerNorm over K folded in, row statistics accumulated beside the MFMAs; LN = 2: ... row statistics emitted by the producer of A
// (rs_in), prefetched into registers before the first tile is requested and combined right before the epilogue
// KS = 2 (round 3): 512 threads = TWO groups of four waves, each group a complete copy of the 2x2 wave layout with its own LDS tiles; group g takes the
// k-tiles g, g + 2, g + 4, ... and the two partial accumulators are added through LDS before the (unchanged) epilogue, which group 0 runs.  A launch of
// the UNet pass lasts as long as one workgroup, and a workgroup's k loop is bound by what ONE wave per SIMD can issue per k-tile (4 x 1 KiB DMA requests
// at ~100+ cycles each, 8 ds_read, 8 MFMA: ~700 cycles for 136 cycles of matrix work at 64 x 64); with two waves per SIMD on different k-tiles the loop
// has half the steps and the SIMD always has a second instruction stream to issue from.  The fp32 sum is (even tiles) + (odd tiles): same value class as
// a 2-way split-K, not the bits of the KS = 1 kernel.
// WGN (round 6): the 4 math waves as 2 x 2 (WGN = 2: every tile of rounds 2-5) or 4 x 1 (WGN = 1: each wave owns BM / 4 rows x ALL BN columns -- the 80-column
// tiles, whose halves are no multiple of 16, and the 160-column tile with an even number of 16-column blocks per wave, which the GEGLU epilogue's value / gate
// pairing needs).  BN need not be a multiple of 32: the B stage is padded to whole rounds of the four waves' 8-row pieces, the pad rows are never requested
// (out-of-range offsets: zero fill) and never read.
// (round 6, measured and NOT kept: one more wave per workgroup that touched both operands 8 k-tiles ahead of the ring -- one 4-byte request per 128-byte tile row,
// its own request queue, in step with the others at the barrier of every k-tile.  The idea: inside a pass every weight comes from HBM and the rings keep only ~4 MB
// of UNIQUE bytes in flight over the chip.  Result on cold operands, every hot shape, every tile: the k loop got 15-50 % SLOWER and the first tile arrived 3-5 us
// later -- the extra requests queue in front of the ring's own, profiles/r06_gemm_kloop_probe_prefetch_wave.txt.)
// WQ = 1 (round 6, W8A16 with the codes resident: osg_gemm_w8.hip): Bt holds uint8 CODES [N][K].  A B tile row is 64 bytes = 4 x 16-byte chunks, one 1-KiB wave-load
// covers 16 rows (lane l: row l >> 2, LDS slot l & 3), chunk c of row r lives at slot c ^ ((r >> 2) & 3) -- rows r, r + 4, r + 8, r + 12 share their banks at a 64-byte
// pitch, the XOR spreads them over the four slots: the 32 lanes of one half of a ds_read_b64 (16 rows x the two 8-byte halves of one chunk) touch every bank
// exactly once.  A lane's fragment of a 32-deep half is ONE ds_read_b64 (8 codes: k = 32 ks + 8 (lane >> 4) .., the same k the A fragment holds) and 8 VALU
// operations (osg_gemm_common.h w8_frag); the B stage is half as large, half the DMA requests, half the bytes through the LDS port per k-tile.  The accumulators
// are scaled once, before any epilogue (w8_scale_acc): split-K slabs, GEGLU, statistics sinks see finished f32 values.  Not with LN (gamma folds into f16 weights).
template <int BM, int BN, int NST, bool CONV, int MODE = 0, int SPEC = 0, int LN = 0, int NCH = 5, int KS = 1, int WGN = 2, int WQ = 0>
__global__ __launch_bounds__((SPEC || KS == 2) ? 512 : 256) void gemm2_kernel(GemmParams pk) {
    // the fields the first DMA request depends on, in ONE batch of scalar loads (GemmParams); everything below reads the register copy
    GemmParams p = pk;
#if OSG_GEMM_PIN
    OSG_PIN(p.A); OSG_PIN(p.Bt); OSG_PIN(p.kdbg); OSG_PIN(p.lda); OSG_PIN(p.strideA); OSG_PIN(p.strideB); OSG_PIN(p.M); OSG_PIN(p.N); OSG_PIN(p.K);
    OSG_PIN(p.splits); OSG_PIN(p.k_per_split); OSG_PIN(p.a_bytes); OSG_PIN(p.b_bytes); OSG_PIN(p.mt); OSG_PIN(p.nt); OSG_PIN(p.n_major); OSG_PIN(p.grid);
    if constexpr (CONV) {
        OSG_PIN(p.H); OSG_PIN(p.W); OSG_PIN(p.Cin); OSG_PIN(p.Ho); OSG_PIN(p.Wo); OSG_PIN(p.KW); OSG_PIN(p.sh); OSG_PIN(p.sw); OSG_PIN(p.pt); OSG_PIN(p.pl);
    }
    OSG_PIN(p.bias); OSG_PIN(p.residual); OSG_PIN(p.rowbias); OSG_PIN(p.rb_ld); OSG_PIN(p.strideC); OSG_PIN(p.ln_c1); OSG_PIN(p.rs_in); OSG_PIN(p.rb_rows);
    OSG_PIN(p.bias_f32); OSG_PIN(p.act); OSG_PIN(p.no_epre); OSG_PIN(p.rs_np);
    if constexpr (WQ) { OSG_PIN(p.wq_sc); OSG_PIN(p.wq_zp); OSG_PIN(p.w_zp); }
#endif
    static_assert(KS == 1 || (KS == 2 && !SPEC && MODE == 0 && LN != 1), "KS = 2: plain kernel only (row statistics come from the producer, LN = 2, or not at all)");
    static_assert(WGN == 1 || WGN == 2, "wave grid: 2 x 2 or 4 x 1");
    static_assert(!WQ || LN == 0, "W8: no folded LayerNorm");
    constexpr int ROWB = 128;                       // bytes per tile row (BK = 64 halves)
    constexpr int ROWB_B = WQ ? 64 : 128;           // ... of the B tile (WQ: 64 codes)
    constexpr int BRW = WQ ? 16 : 8;                // B rows one 1-KiB wave-load covers
    constexpr int BNP = (BN + 4 * BRW - 1) / (4 * BRW) * (4 * BRW);   // B rows of a stage (padded to whole rounds of the four waves' pieces)
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BNP * ROWB_B, GSTAGE = A_BYTES + B_BYTES, STAGE = KS * GSTAGE;
    constexpr int A_LD = BM / 32, B_LD = BNP / (4 * BRW);  // 1-KiB wave-loads per wave per k-tile
    constexpr int WM = BM / (4 / WGN), WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    static_assert(BM % 32 == 0 && WM % 16 == 0 && WN % 16 == 0, "a wave's part of the tile is made of whole 16 x 16 blocks");
    constexpr int INFLIGHT = (NST - 2) * (A_LD + B_LD);
    constexpr unsigned OOB = 0x80000000u;
    static_assert(INFLIGHT <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem2[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    kdbg_stamp(p, 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = SPEC && wave8 >= 4, math = !SPEC || wave8 < 4;
    const bool loads = !SPEC || loader;
    const int wave = wave8 & 3;
    const int grp = KS == 2 ? (wave8 >> 2) : 0;     // k-tile parity this wave works on
    const int wm0 = (WGN == 2 ? (wave >> 1) : wave) * WM;
    const int wn0 = (WGN == 2 ? (wave & 1) : 0) * WN;

    // ---- XCD-aware bijective remap of the flat grid -------------------------------------------------------------------
    const int total = p.grid;                       // (= gridDim.x, which would be one more round trip to the kernel-argument segment)
    int L;
    {
        const int bid = blockIdx.x, x = bid & 7, i = bid >> 3, q = total >> 3, r = total & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int per_batch = p.mt * p.nt * p.splits;
    int zb, m_tile, n_tile, zs;
    {
        zb = L / (per_batch);
        int rem = L - zb * per_batch;
        if (p.n_major) {
            n_tile = rem / (p.splits * p.mt); rem -= n_tile * p.splits * p.mt;
            zs = rem / (p.mt); m_tile = rem - zs * p.mt;
        } else {
            m_tile = rem / (p.splits * p.nt); rem -= m_tile * p.splits * p.nt;
            zs = rem / (p.nt); n_tile = rem - zs * p.nt;
        }
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int kbeg = zs * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg) >> 6;
    const int nsteps = (nkt + KS - 1) / KS;         // (KS = 2: group 1 may run one dummy, zero-filled tile at the end)

    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)zb * p.strideA), 0, p.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bt + (long)zb * p.strideB), 0, p.b_bytes, 0x00020000);

    // ---- per-lane source addressing (constant over the k loop) -------------------------------------------------------------
    const int rsub = lane >> 3;                     // row inside the 8-row group one wave-load covers
    const int gch = (lane & 7) ^ rsub;              // global chunk this lane fetches into LDS slot (lane & 7)
    int a_base[A_LD];                               // byte offset (conv: of tap (0,0); may be negative)
    int a_hi0[A_LD], a_wi0[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; j++) {
        const int m = m0 + (j * 4 + wave) * 8 + rsub;
        if (CONV) {
            const int mm = m < p.M ? m : 0;
            const int hw = p.Ho * p.Wo;
            const int n_img = mm / hw;
            const int r2 = mm - n_img * hw;
            const int ho = r2 / p.Wo, wo = r2 - ho * p.Wo;
            a_hi0[j] = m < p.M ? ho * p.sh - p.pt : -0x40000000;
            a_wi0[j] = wo * p.sw - p.pl;
            a_base[j] = (((n_img * p.H + (ho * p.sh - p.pt)) * p.W + (wo * p.sw - p.pl)) * p.Cin + gch * 8) * 2;
        } else {
            a_base[j] = m < p.M ? (int)(((long)m * p.lda + gch * 8) * 2) : (int)OOB;
            a_hi0[j] = a_wi0[j] = 0;
        }
    }
    int b_base[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; j++) {
        if constexpr (WQ) {
            const int nl = (j * 4 + wave) * 16 + (lane >> 2), n = n0 + nl;
            b_base[j] = (n < p.N && (BNP == BN || nl < BN)) ? (int)((long)n * p.K + (((lane & 3) ^ ((lane >> 4) & 3)) << 4)) : (int)OOB;
        } else {
            const int nl = (j * 4 + wave) * 8 + rsub, n = n0 + nl;
            b_base[j] = (n < p.N && (BNP == BN || nl < BN)) ? (int)(((long)n * p.K + gch * 8) * 2) : (int)OOB;
        }
    }

    // running position of the NEXT tile to issue (conv: decomposed into tap + channel offset, updated incrementally)
    int ik = kbeg + 64 * grp, i_c0 = 0, i_kh = 0, i_kw = 0;
    if (CONV) {
        const int cell = ik / p.Cin;
        i_c0 = ik - cell * p.Cin;
        i_kh = cell / p.KW;
        i_kw = cell - i_kh * p.KW;
    }
    auto issue_tile = [&](int stage) {
        char* As = smem2 + stage * STAGE + grp * GSTAGE;
        char* Bs = As + A_BYTES;
        const bool live = ik < kend;
        const unsigned kill = live ? 0u : OOB;      // past the last k-tile: dummy (zero-filling) loads keep vmcnt uniform
        if (CONV) {
            const int tap_off = ((i_kh * p.W + i_kw) * p.Cin + i_c0) * 2;
#pragma unroll
            for (int j = 0; j < A_LD; j++) {
                const int hi = a_hi0[j] + i_kh, wi = a_wi0[j] + i_kw;
                const bool ok = live && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const unsigned off = ok ? (unsigned)(a_base[j] + tap_off) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(As + (j * 4 + wave) * 1024), 16, off, 0, 0, 0);
            }
#pragma unroll
            for (int adv = 0; adv < KS; adv++) {
                i_c0 += 64;
                if (i_c0 >= p.Cin) { i_c0 = 0; if (++i_kw == p.KW) { i_kw = 0; ++i_kh; } }
            }
        } else {
#pragma unroll
            for (int j = 0; j < A_LD; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(As + (j * 4 + wave) * 1024), 16, (unsigned)a_base[j] | kill, ik * 2, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LD; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(Bs + (j * 4 + wave) * 1024), 16, (unsigned)b_base[j] | kill, WQ ? ik : ik * 2, 0, 0);
        ik += 64 * KS;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read addressing: row = lane & 15, k-chunk = (lane >> 4) (+4 for the second 32-deep half)
    const int frow = lane & 15;
    const int fsw = (((lane >> 4) ^ (frow & 7)) << 4);
    const int a_rd = (wm0 + frow) * ROWB + fsw;
    // (WQ: 8 bytes of chunk 2 ks + (lane >> 5), slot = chunk ^ ((row >> 2) & 3); the second 32-deep half is the same address ^ 32)
    const int b_rd = WQ ? A_BYTES + (wn0 + frow) * ROWB_B + ((((lane >> 5) & 1) ^ ((frow >> 2) & 3)) << 4) + ((lane >> 4) & 1) * 8 : A_BYTES + (wn0 + frow) * ROWB + fsw;

    // LN == 2: the producer's partial row statistics, [M][rs_np][2] floats = rs_np/2 16-byte chunks per row.  Lane l of a wave owns row
    // (l & 15) + 16 ((l >> 4) % TM) of the wave's rows and requests ALL chunks of that row -- BEFORE the first tile (vector-memory
    // results return in order: they are home by the time tile 0 is, at no extra wait).  NCH = chunks per row (template: K <= 64 NCH).
    f32x4 pst[LN == 2 ? NCH : 1];
    if constexpr (LN == 2) {
        const int m = min(m0 + wm0 + (lane & 15) + 16 * ((lane >> 4) % TM), p.M - 1);
        const float* src = p.rs_in + (long)m * p.rs_np * 2;
        const int nch = p.rs_np >> 1;
#pragma unroll
        for (int c = 0; c < NCH; c++)   // unconditional, clamped (a predicated load makes the compiler wait for it on the spot); masked when consumed
            pst[c] = *reinterpret_cast<const f32x4*>(src + 4 * min(c, nch - 1));
        asm volatile("" ::: "memory");
    }

    // epilogue operands of this wave's outputs: requested now, home by the end of the k loop (osg_gemm_common.h epi_prefetch).  They are OLDER than
    // every tile load in the wave's in-order vector-memory queue, so the counted waits of the loop cover them.
    constexpr bool EPRE = TM * TN <= 8;    // (64x64 / 128x64 / 64x128 tiles; the 128x128 tile keeps its on-demand loads: no registers to spare)
    W8Ops<WQ ? TN : 1, true> w8;
    if constexpr (WQ) { if (math) w8_prefetch<TN, true>(p, w8, n0, wn0, lane); }
    EpiOps<TM, TN, CONV, EPRE> epre;
    epre.have = false;
    if (math && !p.ln_c1 && grp == 0) epi_prefetch<TM, TN, CONV, EPRE>(p, epre, m0, n0, wm0, wn0, lane, zb);
    if (loads) {
#pragma unroll
        for (int s2 = 0; s2 < NST - 1; s2++) issue_tile(s2);
    }
    kdbg_stamp(p, 1);
    if constexpr (WQ) { if (math) w8_finalize<TN, true>(w8); }

    float ls[TM], lq[TM];          // LN: running row sums / sums of squares of this lane's rows (see ln_accumulate)
#pragma unroll
    for (int i = 0; i < TM; i++) ls[i] = lq[i] = 0.f;

    int cur = 0, nxt = NST - 1;
    // (round 6, measured and NOT kept: this loop software-pipelined across its barrier -- the MFMAs of tile j - 1 issued between the fragment reads of tile j and the DMA
    // requests of tile j + NST - 1, a second fragment register set, chunk order pinned with sched_barrier; same DMA schedule and summation order, same bits.  -18 % per
    // k-tile when every workgroup reads the SAME tile (634 -> 517 cycles at 64 x 64, 1477 -> 1200 at 128 x 128), nothing on cold operands, nothing in the SD 1.5 pass,
    // +2 % on SDXL and on 4 prompts per GPU: profiles/r06_gemm_kloop_pipelined_*.txt, r06_gemm_kloop_ideal_memory*.txt.  What bounds the loop is the LDS port, shared by
    // the DMA's writes and the fragment reads: profiles/r06_lds_port_probe.txt.)
    for (int kt = 0; kt < nsteps; kt++) {
        if (loads) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // my share of tile kt has landed
        __builtin_amdgcn_s_barrier();                                      // everyone's has; tile kt-1's buffer is free
        if (kt == 0) kdbg_stamp(p, 2);
        if (loads) issue_tile(nxt);
        const char* St = smem2 + cur * STAGE + grp * GSTAGE;
        // (round 3: a pinned order -- half 0's MFMAs with half 1's fragment reads between them, half 1's with the next tile's DMA requests -- was measured
        // against hipcc's own "all reads + requests, wait, all MFMAs": no difference in the k loop of any shape, profiles/r03_interleave_ab.txt; not kept)
        if (math)
#pragma unroll
        for (int ks = 0; ks < (MODE == 1 ? 0 : 2); ks++) {
            f16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const f16x8*>(St + ((a_rd + i * 16 * ROWB) ^ (ks << 6)));
#pragma unroll
            for (int j = 0; j < TN; j++) {
                if constexpr (WQ) b[j] = w8_frag(*reinterpret_cast<const u32x2v*>(St + ((b_rd + j * 16 * ROWB_B) ^ (ks << 5))), w8.zz[j]);
                else b[j] = *reinterpret_cast<const f16x8*>(St + ((b_rd + j * 16 * ROWB) ^ (ks << 6)));
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
            if constexpr (LN == 1) ln_accumulate<TM>(a, ls, lq);
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    kdbg_stamp(p, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the dummy tail loads before the LDS is released
    if (loader) return;
    if constexpr (WQ) w8_scale_acc<TM, TN>(w8, acc);
    if constexpr (KS == 2) {
        // group 1 hands its partial accumulators (and, LN = 1, its partial row sums) to group 0 through the LDS the ring no longer needs
        __builtin_amdgcn_s_barrier();                    // every wave is done reading tiles, no DMA is in flight
        f32x4* red = reinterpret_cast<f32x4*>(smem2) + (wave * (TM * TN + 1)) * 64 + lane;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) red[(i * TN + j) * 64] = acc[i][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a bare s_barrier does not wait for the LDS writes above: osg_tchain.hip lds_barrier)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] += red[(i * TN + j) * 64];
    }
    if constexpr (LN == 2) {
        const int nch = p.rs_np >> 1;
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            // (the empty asm pins the additions AFTER the k loop: hoisted, they would wait for the prefetch before the first tile request)
            asm volatile("" : "+v"(pst[c]));
            if (c < nch) { S += pst[c][0] + pst[c][2]; Q += pst[c][1] + pst[c][3]; }
        }
#pragma unroll
        for (int i = 0; i < TM; i++) {   // row (lane & 15) + 16 i lives in lane (lane & 15) + 16 i
            ls[i] = __shfl(S, (lane & 15) + 16 * i, 64);
            lq[i] = __shfl(Q, (lane & 15) + 16 * i, 64);
        }
    }
    if constexpr (LN != 0) ln_apply<TM, TN, LN == 1>(p, acc, ls, lq, n0, wn0, lane);
    if (OSG_UNLIKELY_IF(!SPEC, p.act == OSG_ACT_GEGLU)) {
        if constexpr (TN % 2 == 0) gemm_epilogue_geglu<TM, TN>(p, acc, m0, n0, wm0, wn0, lane, zb);
        return;
    }
    kdbg_stamp(p, 4);
    float* stat_lds = nullptr;
    if (OSG_UNLIKELY_IF(!SPEC, p.sink[0].table || p.sink[1].table)) {   // (launch_v2 leaves the sinks set only where this epilogue can serve them: one k-slice, 4-aligned shapes)
        __builtin_amdgcn_s_barrier();           // every wave is done with the ring: its first bytes become the waves' staging areas
        stat_lds = reinterpret_cast<float*>(smem2) + wave * (WN * 2);
    }
    if constexpr (KS == 1 && !SPEC && LN == 0 && MODE == 0) {
        if (OSG_UNLIKELY_IF(!SPEC, p.splits > 1 && p.fold_acc)) {
            // split-K, folded by the last workgroup to arrive at the tile (osg_gemm_common.h splitk_fold_acc): it then runs the fused epilogue of an unsplit launch
            if (!splitk_fold_acc<TM, TN>(p, acc, (zb * p.mt + m_tile) * p.nt + n_tile, zs, reinterpret_cast<int*>(smem2), tid)) return;
            EpiOps<TM, TN, CONV, false> none;
            none.have = false;
            gemm_epilogue_fast<TM, TN, CONV, false>(p, acc, m0, n0, wm0, wn0, lane, zb, none, nullptr);
            return;
        }
    }
    gemm_epilogue<TM, TN, CONV, EPRE, !SPEC>(p, acc, m0, n0, wm0, wn0, lane, zb, zb * p.splits + zs, epre, stat_lds);   // (SPEC: 512 threads, 256 registers per lane -- the on-demand operands one block at a time)
    kdbg_stamp(p, 5);
    if (OSG_UNLIKELY_IF(!SPEC, p.kdbg != nullptr)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); kdbg_stamp(p, 6); }
}

template <int BM, int BN, int NST, bool CONV, int MODE = 0, int SPEC = 0, int LN = 0, int NCH = 5, int KS = 1, int WGN = 2, int WQ = 0>
int launch_v2(osg_ctx* ctx, GemmParams& p, int batch) {
    constexpr size_t smem = WQ ? (size_t)NST * KS * (BM * 128 + (BN + 63) / 64 * 64 * 64) : (size_t)NST * KS * (BM + (BN + 31) / 32 * 32) * 128;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static_assert(KS == 1 || (size_t)4 * ((BM / 32) * (BN / 32) + 1) * 1024 <= smem, "KS = 2: the accumulator hand-over must fit the ring");
    auto kern = gemm2_kernel<BM, BN, NST, CONV, MODE, SPEC, LN, NCH, KS, WGN, WQ>;
    static unsigned long long attr_mask = 0;   // (per device: hipFuncSetAttribute is, and a process may hold several)
    if (osg_first_on_device(attr_mask)) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    p.mt = (p.M + BM - 1) / BM;
    p.nt = (p.N + BN - 1) / BN;
    if (KS == 2 || SPEC || LN != 0 || MODE != 0) p.fold_acc = 0;   // (the in-kernel split-K fold is the plain kernel's: a 256-thread protocol)
    dim3 grid((unsigned)(p.mt * p.nt * batch * p.splits));
    p.grid = (int)grid.x;
    p.no_epre = osg_mm::no_epi_prefetch();
    p.kdbg = kdbg_buffer(ctx, grid.x);
    const osg_mm::StatSink sinks_in[2] = {p.sink[0], p.sink[1]};     // (p is the caller's: a reduce launch that follows still wants them)
    if (p.sink[0].table || p.sink[1].table) {
        // GroupNorm statistics from this launch's epilogue (StatSink): only the real launch of a pass (not the tuner's repetitions), one k-slice, the compact
        // epilogue, whole tiles inside one image; otherwise the caller's follow-up launch computes them (osg_conv2d_nhwc_v)
        const bool ok = !ctx->tuning && p.splits == 1 && batch == 1 && MODE == 0 && LN == 0 && !SPEC && p.act != OSG_ACT_GEGLU && (p.N & 3) == 0 && ((p.ldc | p.ldc2) & 3) == 0 &&
                        p.sink_hw > 0 && p.sink_hw % BM == 0 && p.M % p.sink_hw == 0;
        if (ok) { ctx->sink_fused = true; p.sink_imgs = p.M / p.sink_hw; p.sink_per_xcd = ctx->xcd_ids8 ? 1 : 0; }
        else p.sink[0].table = p.sink[1].table = nullptr;
    }
    hipLaunchKernelGGL(kern, grid, dim3((SPEC || KS == 2) ? 512 : 256), smem, ctx->compute, p);
    p.sink[0] = sinks_in[0]; p.sink[1] = sinks_in[1];
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // namespace
