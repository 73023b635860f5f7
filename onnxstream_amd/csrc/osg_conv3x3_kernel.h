// libosgpu: the halo-reuse 3x3 convolution kernel and its launcher (conv3x3_kernel, launch3), shared by osg_conv3x3.hip (f16 weights) and osg_conv3x3_w8.hip
// (uint8 weight codes resident, WQ = 1).  The description is at the top of osg_conv3x3.hip.
#pragma once
#include "osg_gemm_common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

using namespace osg_mm;

namespace {

template <int I> using ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- compile-time geometry of one instantiation ------------------------------------------------------------------------------
// WQ = 1 (round 6, W8A16 with the codes resident): the weight tile is [BN][64] uint8 CODES -- 64-byte rows, 16 rows per 1-KiB wave-load, chunk c of row r at slot
// c ^ ((r >> 2) & 3) -- read as one ds_read_b64 per fragment half and turned into halves between LDS and the MFMA (osg_gemm_common.h w8_frag; gemm2_kernel WQ has the
// full description); stages are half as large, the ring holds up to 8 of them.
template <int W_, int BN, int NLW = 4, int WQ = 0>
struct Geo {
    static constexpr int TI = W_ == 8 ? 2 : 1;              // images per tile
    static constexpr int TH = 128 / (W_ * TI);              // output rows per image per tile
    static constexpr int PW = W_ == 8 ? 16 : W_ + 2;        // patch row pitch in pixels (16 for W=8 keeps fragment reads conflict-free)
    static constexpr int PH = TH + 2;
    static constexpr int PPI = PH * PW;                     // patch pixels per image
    static constexpr int PP = TI * PPI;
    static constexpr int NPW = ((PP + 7) / 8 + NLW - 1) / NLW;   // patch wave-loads per LOADER wave (NLW of them: 4, or 8 = 768-thread workgroups)
    static constexpr int PATCH_BYTES = NPW * NLW * 1024;
    static constexpr int BRW = WQ ? 16 : 8;                 // weight rows one 1-KiB wave-load covers
    static constexpr int WLB = ((BN + BRW - 1) / BRW + NLW - 1) / NLW;    // weight wave-loads per loader wave per tap
    static constexpr int BST_BYTES = WLB * NLW * 1024;      // one weight stage (padded to whole rounds of the loader waves)
    static constexpr int NSTW_ = (160 * 1024 - 2 * PATCH_BYTES) / BST_BYTES;
    static constexpr int NSTW = NSTW_ > 8 ? 8 : NSTW_;      // weight stages: whatever the 160 KiB of LDS hold (the kernel is
    static constexpr int D = NSTW - 1;                      // latency-bound: bytes in flight per CU are what buys throughput)
    static constexpr int PT = 10 - D;                       // the next slab's patch is issued during taps 0..PT-1 ...
    static constexpr int PPT = (NPW + PT - 1) / PT;         // ... PPT pieces per tap per wave
    static constexpr size_t SMEM = 2 * (size_t)PATCH_BYTES + (size_t)NSTW * BST_BYTES;
    static constexpr bool OK = D >= 3 && PT >= 2;          // (the pipeline needs >= 3 units of weights in flight and >= 2 taps to spread the next patch over)
    static constexpr int patch_loads(int t) {               // real patch pieces issued at tap t
        if (t >= PT) return 0;
        int n = NPW - t * PPT;
        return n < 0 ? 0 : (n > PPT ? PPT : n);
    }
    static constexpr int allowed_outstanding(int t) {       // at the top of tap t: what the previous D-2 units issued
        int n = 0;
        for (int d = 1; d <= D - 2; d++) n += WLB + patch_loads((t - d + 18) % 9);
        return n;
    }
};

// W_: image width (= tile width); BN: output channels per tile; WGM x WGN: grid of the 4 MATH waves.
// 512 threads = 4 math waves + 4 LOADER waves (one pair per SIMD).  Measured (tools/pmc_conv.sh): one `buffer_load ... lds`
// costs its issuing wave ~100+ cycles, so with loads and MFMAs in the same instruction stream the matrix pipe idled 75 % of
// the time whatever the ring depth; with the DMA issue moved to waves that do nothing else, the math waves' stream is
// ds_read + MFMA only and the two streams overlap on the SIMD.
// (the body is a __device__ function: hipcc emits no host stub for a __global__ template whose body holds a generic lambda)
template <int W_, int BN, int WGM, int WGN, int MODE, int NLW, int WQ = 0>
__device__ __forceinline__ void conv3x3_body(const GemmParams& p) {
    using G = Geo<W_, BN, NLW, WQ>;
    constexpr int TI = G::TI, TH = G::TH, PW = G::PW, PPI = G::PPI, PP = G::PP, NPW = G::NPW, PATCH_BYTES = G::PATCH_BYTES;
    constexpr int WLB = G::WLB, BST_BYTES = G::BST_BYTES, NSTW = G::NSTW, D = G::D, PPT = G::PPT;
    constexpr int WM = 128 / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && WGM * WGN == 4, "bad wave layout");
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) char smem3[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    char* const patch0 = smem3;                      // 2 patch buffers, then NSTW weight stages
    char* const bst0 = smem3 + 2 * PATCH_BYTES;

    kdbg_stamp(p, 0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = loader ? wave8 - 4 : wave8;     // index inside the role (4 math waves, NLW loader waves)

    // ---- XCD-aware bijective remap of the flat grid (see osg_gemm.hip) ------------------------------------------------
    int L;
    {
        const int total = p.grid, bid = blockIdx.x, x = bid & 7, i = bid >> 3, q = total >> 3, r = total & 7;   // (p.grid = gridDim.x, without the trip to the hidden arguments)
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    int m_tile, n_tile, zs;
    if (p.n_major) {
        n_tile = L / (p.splits * p.mt); L -= n_tile * p.splits * p.mt;
        zs = L / (p.mt); m_tile = L - zs * p.mt;
    } else {
        m_tile = L / (p.splits * p.nt); L -= m_tile * p.splits * p.nt;
        zs = L / (p.nt); n_tile = L - zs * p.nt;
    }
    const int m0 = m_tile * 128, n0 = n_tile * BN;
    const int slab_b = zs * (p.k_per_split >> 6);
    const int slab_e = min(p.Cin >> 6, slab_b + (p.k_per_split >> 6));

    if (loader) {
        // =========================================== LOADER waves ==============================================================
        const int img0 = m0 / (p.H * W_);                       // first image of the tile
        const int y0 = (m0 - img0 * p.H * W_) / W_;             // first output row (0 when the tile holds whole images)
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bt, 0, p.b_bytes, 0x00020000);
        const int rsub = lane >> 3;
        const int gch = (lane & 7) ^ rsub;
        // patch source offsets: this wave issues patch wave-loads g = piece * 4 + wave, piece = 0..NPW-1
        unsigned pa_off[NPW];
#pragma unroll
        for (int pc = 0; pc < NPW; pc++) {
            const int pp = (pc * NLW + wave) * 8 + rsub;        // patch pixel
            const int ti = pp / PPI, rr = pp - ti * PPI;
            const int py = rr / PW, px = rr - py * PW;
            const int img = img0 + ti, y = y0 + py - 1, x = px - 1;
            const bool ok = pp < PP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)W_ && img * p.H * W_ < p.M;
            pa_off[pc] = ok ? (unsigned)((((img * p.H + y) * W_ + x) * p.Cin + gch * 8) * 2) : OOB;
        }
        // weight source offsets: wave-load j covers rows (j*4 + wave)*8 .. +7 of the [BN][64] tile
        unsigned b_off[WLB];
#pragma unroll
        for (int j = 0; j < WLB; j++) {
            if constexpr (WQ) {
                const int nl = (j * NLW + wave) * 16 + (lane >> 2);
                const int n = n0 + nl;
                b_off[j] = (nl < BN && n < p.N) ? (unsigned)((long)n * p.K + (((lane & 3) ^ ((lane >> 4) & 3)) << 4)) : OOB;
            } else {
                const int nl = (j * NLW + wave) * 8 + rsub;
                const int n = n0 + nl;
                b_off[j] = (nl < BN && n < p.N) ? (unsigned)(((long)n * p.K + gch * 8) * 2) : OOB;
            }
        }
        auto issue_patch_piece = [&](int pc, int slab, char* buf) {     // slab may be past the end: dummy (zero-filling) load
            if (MODE >= 3 && MODE != 6) return;
            const unsigned kill = (slab < slab_e && MODE != 2) ? 0u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(buf + (pc * NLW + wave) * 1024), 16, pa_off[pc] | kill, slab * 128, 0, 0);
        };
        auto issue_weights = [&](int stage, int tap, int slab) {
            if (MODE >= 3 && MODE != 6) return;
            const unsigned kill = (slab < slab_e && MODE != 2) ? 0u : OOB;
            const int soff = (tap * p.Cin + slab * 64) * (WQ ? 1 : 2);
            char* dst = bst0 + stage * BST_BYTES;
#pragma unroll
            for (int j = 0; j < WLB; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(dst + (j * NLW + wave) * 1024), 16, b_off[j] | kill, soff, 0, 0);
        };
        // prologue == units -D .. -1 of the steady state (D = NSTW - 1 units of weights in flight)
        issue_weights(0, 0, slab_b);
#pragma unroll
        for (int pc = 0; pc < NPW; pc++) issue_patch_piece(pc, slab_b, patch0);
#pragma unroll
        for (int d = 1; d < D; d++) issue_weights(d, d % 9, slab_b + d / 9);
        wait_vmcnt<(D - 1) * WLB>();                         // unit 0's weights + the first patch have landed
        __builtin_amdgcn_s_barrier();
        int ust = 0;                                         // weight stage of the current unit
        for (int slab = slab_b; slab < slab_e; slab++) {
            char* patch_next = patch0 + (((slab - slab_b) & 1) ^ 1) * PATCH_BYTES;
            static_for<0, 9>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                // after this wait + barrier the tiles of unit u+1 are resident too: what the previous D-2 units issued may stay in flight
                wait_vmcnt<G::allowed_outstanding(t)>();
                if (MODE != 5) __builtin_amdgcn_s_barrier();
                constexpr int td = (t + D) % 9;
                int std_ = ust + D;
                std_ = std_ >= NSTW ? std_ - NSTW : std_;
                issue_weights(std_, td, slab + (t + D) / 9);   // into the stage unit u-1 just released
                static_for<0, G::patch_loads(t)>([&](auto pc) { issue_patch_piece(t * PPT + decltype(pc)::value, slab + 1, patch_next); });
                ust = ust + 1 == NSTW ? 0 : ust + 1;
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the dummy tail loads before the LDS is released
        return;
    }

    // ============================================== MATH waves ==================================================================
    const int wm0 = (wave / WGN) * WM;
    const int wn0 = (wave % WGN) * WN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addressing
    const int frow = lane & 15, fq = lane >> 4;
    int pp0[TM];                                        // patch pixel of this lane's output pixel at tap (0,0)
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int ml = wm0 + i * 16 + frow;
        const int ti = ml / (TH * W_), rr = ml - ti * (TH * W_);
        const int ty = rr / W_, tx = rr - ty * W_;
        pp0[i] = ti * PPI + ty * PW + tx;
    }
    const int b_rd = WQ ? (wn0 + frow) * 64 + ((((lane >> 5) & 1) ^ ((frow >> 2) & 3)) << 4) + ((lane >> 4) & 1) * 8 : (wn0 + frow) * 128 + ((fq ^ (frow & 7)) << 4);
    constexpr bool W8SC = false;           // (no registers for 4 TN scales through the k loop; a convolution's weight has ONE scale -- a kernel argument -- unless the caller hands vectors)
    W8Ops<WQ ? TN : 1, W8SC> w8;
    if constexpr (WQ) w8_prefetch<TN, W8SC>(p, w8, n0, wn0, lane);

    // fragment sets: set 0 holds (unit, k-half 0), set 1 (unit, k-half 1); each is fetched while the MFMAs of the other run, the
    // first half of unit u+1 already during the second half of unit u (its tiles are guaranteed resident one barrier ahead).
    f16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto read_frags = [&](f16x8 (&fa)[TM], f16x8 (&fb)[TN], const char* pbuf, const char* bbuf, int tap_off, int ks) {
        if (MODE == 1 || (MODE == 4 && ks == 1)) return;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int pp = pp0[i] + tap_off;
            fa[i] = *reinterpret_cast<const f16x8*>(pbuf + ((pp * 128 + ((fq ^ (pp & 7)) << 4)) ^ (ks << 6)));
        }
#pragma unroll
        for (int j = 0; j < TN; j++) {
            if constexpr (WQ) fb[j] = w8_frag(*reinterpret_cast<const u32x2v*>(bbuf + ((b_rd + j * 16 * 64) ^ (ks << 5))), w8.zz[j]);
            else fb[j] = *reinterpret_cast<const f16x8*>(bbuf + ((b_rd + j * 16 * 128) ^ (ks << 6)));
        }
    };
    auto mma = [&](f16x8 (&fa)[TM], f16x8 (&fb)[TN]) {
        if (MODE == 1) return;   // experiment: loads + barriers only
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    };

    // WQ: the B fragments in three steps -- codes out of LDS (rb0 / rb1, two halves ahead), codes -> halves (8 VALU operations per fragment, issued between the
    // MFMAs of the half in front), MFMA: see the loop
    u32x2v rb0[WQ ? TN : 1], rb1[WQ ? TN : 1];
    auto read_a = [&](f16x8 (&fa)[TM], const char* pbuf, int tap_off, int ks) {
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int pp = pp0[i] + tap_off;
            fa[i] = *reinterpret_cast<const f16x8*>(pbuf + ((pp * 128 + ((fq ^ (pp & 7)) << 4)) ^ (ks << 6)));
        }
    };
    auto read_codes = [&](u32x2v (&rb)[WQ ? TN : 1], const char* bbuf, int ks) {
        if constexpr (WQ) {
#pragma unroll
            for (int j = 0; j < TN; j++) rb[j] = *reinterpret_cast<const u32x2v*>(bbuf + ((b_rd + j * 16 * 64) ^ (ks << 5)));
        }
    };
    auto convert = [&](f16x8 (&fb)[TN], const u32x2v (&rb)[WQ ? TN : 1]) {
        if constexpr (WQ) {
#pragma unroll
            for (int j = 0; j < TN; j++) fb[j] = w8_frag(rb[j], w8.zz[j]);
        }
    };

    // the epilogue's operands (bias, per-image bias, residual) of this wave's outputs: requested now -- the math waves have no other vector-memory
    // traffic -- and home long before the last tap (osg_gemm_common.h epi_prefetch)
    constexpr bool EPRE = TM * TN <= 10 && NLW == 4;   // (the 128x80 tile; the wider tiles and the 768-thread variant have no registers to spare)
    EpiOps<TM, TN, true, EPRE> epre;
    epi_prefetch<TM, TN, true, EPRE>(p, epre, m0, n0, wm0, wn0, lane, 0);
    kdbg_stamp(p, 1);
    __builtin_amdgcn_s_barrier();                       // unit 0's weights + the first patch have landed
    kdbg_stamp(p, 2);
    if constexpr (WQ) {
        w8_finalize<TN, W8SC>(w8);
        read_a(fa0, patch0, 0, 0);
        read_codes(rb0, bst0, 0);
        read_codes(rb1, bst0, 1);
        convert(fb0, rb0);
    } else
    read_frags(fa0, fb0, patch0, bst0, 0, 0);
    if (MODE == 4) {
#pragma unroll
        for (int i = 0; i < TM; i++) fa1[i] = fa0[i];
#pragma unroll
        for (int j = 0; j < TN; j++) fb1[j] = fb0[j];
    }
    int ust = 0;
    for (int slab = slab_b; slab < slab_e; slab++) {
        const char* patch = patch0 + ((slab - slab_b) & 1) * PATCH_BYTES;
        const char* patch_next = patch0 + (((slab - slab_b) & 1) ^ 1) * PATCH_BYTES;
        static_for<0, 9>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (MODE != 5) __builtin_amdgcn_s_barrier();               // units <= u+1 resident; the loaders may now overwrite unit u-1's stage
            constexpr int kh = t / 3, kw = t % 3;
            const char* Bs = bst0 + ust * BST_BYTES;
            // sched_barrier(0) pins the order "issue the NEXT half's ds_reads, then run THIS half's MFMAs": left alone, hipcc sinks
            // the reads next to their uses and re-serialises LDS latency with the matrix pipe
            if constexpr (WQ) {
                // unit u, half 0's MFMAs: A fragments of (u, 1), codes of (u + 1, 0), conversion of (u, 1)'s codes (read one half ago: landed) between them;
                // half 1's MFMAs: A fragments of (u + 1, 0), codes of (u + 1, 1), conversion of (u + 1, 0)'s.  Units <= u + 1 are resident behind this tap's barrier.
                constexpr int KV = (8 * TN + TM * TN - 1) / (TM * TN);   // VALU operations per MFMA that spread the conversions over the half
                const int ustn = ust + 1 == NSTW ? 0 : ust + 1;
                const char* Bn = bst0 + ustn * BST_BYTES;
                read_a(fa1, patch, kh * PW + kw, 1);
                read_codes(rb0, Bn, 0);
                convert(fb1, rb1);
                mma(fa0, fb0);
                static_for<0, TM * TN>([&](auto ic_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (decltype(ic_)::value < TM + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, KV, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
                ust = ustn;
                constexpr int tn8 = (t + 1) % 9;
                read_a(fa0, t == 8 ? patch_next : patch, (tn8 / 3) * PW + tn8 % 3, 0);
                read_codes(rb1, Bn, 1);
                convert(fb0, rb0);
                mma(fa1, fb1);
                static_for<0, TM * TN>([&](auto ic_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (decltype(ic_)::value < TM + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, KV, 0);
                });
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (MODE == 0 || MODE == 7) {
                // round 3 (tools/kernel_phase_probe.py, OSG_CONV3X3_DBG=6 = the old order): the NEXT half's fragment reads interleaved one by one with THIS half's MFMAs (sched_group_barrier) instead of issued in a block
                // in front of them -- an MFMA occupies the matrix pipe for ~16 cycles in which the wave can issue other instructions
                read_frags(fa1, fb1, patch, Bs, kh * PW + kw, 1);
                mma(fa0, fb0);
                if constexpr (MODE == 0) {
                    static_for<0, TM + TN>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); });
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                } else {
                    static_for<0, TM + TN>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); });
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                ust = ust + 1 == NSTW ? 0 : ust + 1;
                constexpr int tn6 = (t + 1) % 9;
                read_frags(fa0, fb0, t == 8 ? patch_next : patch, bst0 + ust * BST_BYTES, (tn6 / 3) * PW + tn6 % 3, 0);
                mma(fa1, fb1);
                if constexpr (MODE == 0) {
                    static_for<0, TM + TN>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); });
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                } else {
                    static_for<0, TM + TN>([&](auto) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); });
                    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
            read_frags(fa1, fb1, patch, Bs, kh * PW + kw, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            ust = ust + 1 == NSTW ? 0 : ust + 1;
            constexpr int tn = (t + 1) % 9;
            if (MODE != 4) read_frags(fa0, fb0, t == 8 ? patch_next : patch, bst0 + ust * BST_BYTES, (tn / 3) * PW + tn % 3, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    kdbg_stamp(p, 3);
    if constexpr (WQ) { if constexpr (W8SC) w8_scale_acc<TM, TN>(w8, acc); else w8_scale_acc_late<TM, TN>(p, acc, n0, wn0, lane); }
    kdbg_stamp(p, 4);
    float* stat_lds = nullptr;
    if (OSG_UNLIKELY(p.sink[0].table || p.sink[1].table)) {   // GroupNorm statistics of this launch's output (osg_gemm_common.h StatSink; launch3 decides)
        __builtin_amdgcn_s_barrier();           // (the loader waves have left: the four math waves) every one is done with patches and weight stages
        stat_lds = reinterpret_cast<float*>(smem3) + (wave8 & 3) * (WN * 2);
    }
    if constexpr (MODE == 0) {
        if (OSG_UNLIKELY(p.splits > 1 && p.fold_acc)) {
            // split-K over slabs, folded by the last workgroup to arrive at the tile (osg_gemm_common.h splitk_fold_acc; only the 4 math waves are still here:
            // tid 0..255): it then runs the fused epilogue of an unsplit launch
            if (!splitk_fold_acc<TM, TN>(p, acc, m_tile * p.nt + n_tile, zs, reinterpret_cast<int*>(smem3), tid)) return;
            EpiOps<TM, TN, true, false> none;
            none.have = false;
            gemm_epilogue_fast<TM, TN, true, false, false>(p, acc, m0, n0, wm0, wn0, lane, 0, none, nullptr);
            return;
        }
    }
    gemm_epilogue<TM, TN, true, EPRE, false>(p, acc, m0, n0, wm0, wn0, lane, 0, zs, epre, stat_lds);
    kdbg_stamp(p, 5);
    if (OSG_UNLIKELY(p.kdbg != nullptr)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); kdbg_stamp(p, 6); }
}

template <int W_, int BN, int WGM, int WGN, int MODE, int NLW, int WQ = 0>
__global__ __launch_bounds__(256 + 64 * NLW) void conv3x3_kernel(GemmParams pk) {
    // the fields the first DMA requests / the epilogue prefetch depend on, in ONE batch of scalar loads at entry (GemmParams, round 6)
    GemmParams p = pk;
    OSG_PIN(p.A); OSG_PIN(p.Bt); OSG_PIN(p.kdbg); OSG_PIN(p.M); OSG_PIN(p.N); OSG_PIN(p.K); OSG_PIN(p.splits); OSG_PIN(p.k_per_split); OSG_PIN(p.a_bytes); OSG_PIN(p.b_bytes);
    OSG_PIN(p.mt); OSG_PIN(p.nt); OSG_PIN(p.n_major); OSG_PIN(p.grid); OSG_PIN(p.H); OSG_PIN(p.Cin);
    OSG_PIN(p.bias); OSG_PIN(p.residual); OSG_PIN(p.rowbias); OSG_PIN(p.rb_ld); OSG_PIN(p.rb_rows); OSG_PIN(p.bias_f32); OSG_PIN(p.act); OSG_PIN(p.no_epre);
    if constexpr (WQ) { OSG_PIN(p.wq_sc); OSG_PIN(p.wq_zp); OSG_PIN(p.w_zp); }
    conv3x3_body<W_, BN, WGM, WGN, MODE, NLW, WQ>(p);
}

template <int W_, int BN, int WGM, int WGN, int MODE = 0, int NLW = 4, int WQ = 0>
int launch3(osg_ctx* ctx, GemmParams& p) {
    constexpr size_t smem = Geo<W_, BN, NLW, WQ>::SMEM;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static_assert(Geo<W_, BN, NLW, WQ>::OK, "pipeline depth out of range");
    auto kern = conv3x3_kernel<W_, BN, WGM, WGN, MODE, NLW, WQ>;
    static unsigned long long attr_mask = 0;   // (per device: hipFuncSetAttribute is, and a process may hold several)
    if (osg_first_on_device(attr_mask)) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    p.mt = (p.M + 127) / 128;
    p.nt = (p.N + BN - 1) / BN;
    if (MODE != 0) p.fold_acc = 0;
    p.no_epre = osg_mm::no_epi_prefetch();
    p.kdbg = kdbg_buffer(ctx, (long)p.mt * p.nt * p.splits);
    const osg_mm::StatSink sinks_in[2] = {p.sink[0], p.sink[1]};
    if (p.sink[0].table || p.sink[1].table) {   // (see launch_v2 in osg_gemm.hip)
        const bool ok = !ctx->tuning && p.splits == 1 && MODE == 0 && (p.N & 3) == 0 && ((p.ldc | p.ldc2) & 3) == 0 && p.sink_hw > 0 && p.sink_hw % 128 == 0 && p.M % p.sink_hw == 0;
        if (ok) { ctx->sink_fused = true; p.sink_imgs = p.M / p.sink_hw; p.sink_per_xcd = ctx->xcd_ids8 ? 1 : 0; }
        else p.sink[0].table = p.sink[1].table = nullptr;
    }
    p.grid = p.mt * p.nt * p.splits;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(256 + 64 * NLW), smem, ctx->compute, p);
    p.sink[0] = sinks_in[0]; p.sink[1] = sinks_in[1];
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // namespace
