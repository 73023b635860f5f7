// The LEAN linear kernel (round 4): y[M,N] = [LayerNorm](x[M,K]) . W^T + bias + residual for the transformer blocks' projections and the 1x1 convolutions
// -- K <= 2560, a few GFLOP each, ~90 launches of the SD 1.5 pass.  (reference: MatMul + Add (+ Add) src/onnxstream.cpp:5669-5861, :3906-4000; the 9-op
// LayerNorm chain :5237-5604; 1x1 Conv :4494-4707 -> XnnPack::convolution :1292)
//
// Why a second GEMM kernel.  These launches do 0.7 us of matrix work and took 12-15 us each inside the pass: gemm2_kernel spends ~2 us between entry and its
// first request (per-lane address tables, the implicit-GEMM decode, a cold instruction cache), then walks its k loop one ring slot at a time against
// operands that are cold at every launch.  What osg_tchain.hip measured carries over: a launch of this size lasts as long as ONE memory
// round trip if -- and only if -- everything it will ever read is requested at entry.  So here
//   * the workgroup's x rows (BM x K, <= 80 KB) go to LDS with one burst of LDS-DMA requests (the k-tiled, XOR-swizzled image of gemm2's A tiles),
//   * every wave requests the first 320 k of ITS weight columns straight into registers (kn8 layout [K/8][N][8], osg_tblock_pack_weight: a fragment
//     request is four runs of 256 contiguous bytes), refilling a slot as soon as it has been read -- no LDS ring, no barrier in the k loop,
//   * bias and residual fragments are requested at entry too, and the prologue is ~60 instructions;
//   * LayerNorm, when asked for, is the standalone kernel's arithmetic on the rows in LDS (the rows are complete: K = C), with the ORIGINAL weights -- no
//     folded copies, no row statistics from the producer.
// Numerics: f32 accumulation in k order on v_mfma_f32_16x16x32_f16, (acc + bias) + residual, one RNE rounding -- gemm2_kernel's unsplit form.
#include "osg_common.h"

#include <cstdlib>
#include <type_traits>

namespace osg_ls {

// Workgroup barrier behind LDS writes.  A bare s_barrier does not wait for the wave's own outstanding LDS operations (gfx90a and later back off at barriers, so
// hipcc inserts no s_waitcnt in front of one), and LDS requests of different SIMDs are not served in issue order: without the wait a wave past the barrier can
// read what another wave has issued but the LDS has not yet written -- seen as run-to-run differences of osg_qattn at 10 heads of 64 on cold operands
// (profiles/r04_qattn_lds_barrier_race.txt).  lgkmcnt only: vector-memory requests (weight fragments in flight) are deliberately NOT waited for.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}


struct LinParams {
    const f16* x; long ldx;
    const f16* w;                 // kn8 [K/8][N][8]
    const f16* bias; const f16* res; long ldr;
    const f16* gamma; const f16* beta; float eps;
    f16* y; long ldy; f16* y2; long ldy2;
    float* rs_out;                // osg_gemm_rowstats' hand-over: [M][N/32][2] = (sum, sum of squares) of the ROUNDED outputs over every 32-column slot (NULL: none)
    int M, N, K, mt, nt;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for_impl(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) { static_for_impl<0, N>(f); }

__device__ __forceinline__ f16x8 ldb(__amdgpu_buffer_rsrc_t rs, unsigned lo, int so) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, so, 0));
}

// LayerNorm of the BM rows of the LDS image, in place: layer_norm_kernel's arithmetic (osg_norm.hip: mean, then the sum of squared deviations, fp32),
// 256 / BM lanes per row.  gamma / beta: LDS copies ([K] f16 each).
template <int BM>
__device__ __forceinline__ void ln_rows_inplace(char* A, const char* gamma, const char* beta, int K, float eps, int tid) {
    constexpr int LPR = 256 / BM, TILE = BM * 128;
    const int row = tid / LPR, part = tid % LPR;
    const int nch = K >> 3;                                      // 16-byte chunks per row
    auto at = [&](int c) { return A + (c >> 3) * TILE + row * 128 + (((c & 7) ^ (row & 7)) << 4); };
    float s = 0.f;
    for (int c = part; c < nch; c += LPR) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(at(c));
#pragma unroll
        for (int e = 0; e < 8; e++) s += (float)t[e];
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)K;
    float q = 0.f;
    for (int c = part; c < nch; c += LPR) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(at(c));
#pragma unroll
        for (int e = 0; e < 8; e++) { const float d = (float)t[e] - mean; q += d * d; }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
    for (int c = part; c < nch; c += LPR) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(at(c));
        const f16x8 gm = *reinterpret_cast<const f16x8*>(gamma + c * 16);
        const f16x8 bt = *reinterpret_cast<const f16x8*>(beta + c * 16);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (f16)(((float)t[e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
        *reinterpret_cast<f16x8*>(at(c)) = o;
    }
}

// BM rows per workgroup; WGM x WGN waves, each (BM / WGM) rows x (16 TN) columns; BN = 16 TN WGN columns per workgroup.  K % 320 == 0.
template <int BM, int WGM, int WGN, int TN, bool LN>
__global__ __launch_bounds__(256) void linear_small_kernel(LinParams p) {
    static_assert(WGM * WGN == 4 && (BM / WGM) % 16 == 0, "four waves");
    constexpr int WM = BM / WGM, TM = WM / 16, WN = 16 * TN, BN = WN * WGN;
    constexpr int TILE = BM * 128;             // bytes of one 64-deep k-tile of the row block
    constexpr int D = 10;                      // 32-deep k steps in flight per wave (320 k)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;
    // XCD-contiguous walk, column tiles outermost: the workgroups of one XCD (workgroup i runs on XCD i mod 8) share a few column tiles' weights in its L2
    int L;
    {
        const int total = gridDim.x, bid = blockIdx.x, x = bid & 7, i = bid >> 3, q = total >> 3, r = total & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int n_blk = L / p.mt, m_blk = L - n_blk * p.mt;
    const int m0 = m_blk * BM, n0 = n_blk * BN;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int K = p.K, N = p.N;
    const int nkt = K >> 6;

    // ---- everything this workgroup will read, requested now ---------------------------------------------------------------------------------
    // (1) the weight fragments of the first D k steps: lane = (k chunk g, column n0 + wn0 + l16 (+ 16 j)); step s adds 4 k chunks
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, K * N * 2, 0x00020000);
    const unsigned lo = (unsigned)((g * N + n0 + wn0 + l16) * 16);
    const int kstep = 4 * N * 16;              // bytes per 32-deep k step
    f16x8 bq[D][TN];
    static_for<D>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int j = 0; j < TN; j++) bq[s][j] = ldb(rsW, lo, s * kstep + j * 256);
    });
    // (2) the row block -> LDS (k-tiled, chunk c of row r at slot c ^ (r & 7): the swizzle on the source side, as in gemm2_kernel)
    {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long)m0 * p.ldx), 0, (int)(((long)(BM - 1) * p.ldx + K) * 2), 0x00020000);
        const int rsub = lane >> 3, gch = (lane & 7) ^ rsub;
        for (int kt = 0; kt < nkt; kt++)
#pragma unroll
            for (int qq = 0; qq < BM / 32; qq++) {
                const int q8 = qq * 4 + wave;                       // 8-row group inside the tile
                const unsigned off = (unsigned)(((long)(q8 * 8 + rsub) * p.ldx + kt * 64 + gch * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(lds + kt * TILE + q8 * 1024), 16, off, 0, 0, 0);
            }
        if constexpr (BM == 16) {                                   // (two 8-row groups: waves 0 and 1 carry them, even / odd k-tiles split with waves 2 and 3)
            for (int kt = wave >> 1; kt < nkt; kt += 2) {
                const int q8 = wave & 1;
                const unsigned off = (unsigned)(((long)(q8 * 8 + rsub) * p.ldx + kt * 64 + gch * 8) * 2);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(lds + kt * TILE + q8 * 1024), 16, off, 0, 0, 0);
            }
        }
    }
    // (3) bias and residual fragments of this wave's outputs
    f16x4 bv[TN], rv[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        bv[j] = f16x4{0, 0, 0, 0};
        if (p.bias) bv[j] = *reinterpret_cast<const f16x4*>(p.bias + n0 + wn0 + j * 16 + g * 4);
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            rv[i][j] = f16x4{0, 0, 0, 0};
            if (p.res) rv[i][j] = *reinterpret_cast<const f16x4*>(p.res + (long)(m0 + wm0 + i * 16 + l16) * p.ldr + n0 + wn0 + j * 16 + g * 4);
        }
    // (4) LayerNorm operands -> LDS, behind the row block
    char* const GB = lds + nkt * TILE;
    if constexpr (LN) {
        for (int c = tid; c < (K >> 3); c += 256) {
            *reinterpret_cast<f16x8*>(GB + c * 16) = *reinterpret_cast<const f16x8*>(p.gamma + c * 8);
            *reinterpret_cast<f16x8*>(GB + K * 2 + c * 16) = *reinterpret_cast<const f16x8*>(p.beta + c * 8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if constexpr (LN) {
        ln_rows_inplace<BM>(lds, GB, GB + K * 2, K, p.eps, tid);
        lds_barrier();
    }

    // ---- the contraction: groups of D k steps; a slot is refilled with the step D further on as soon as its MFMAs are issued ----------------------
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_rd = (wm0 + l16) * 128 + ((g ^ (l16 & 7)) << 4);
    const int ngroups = K / (32 * D);
    for (int grp = 0; grp < ngroups; grp++) {
        const bool more = grp + 1 < ngroups;
        static_for<D>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int step = grp * D + s;                            // 32-deep k step: k-tile step >> 1, half step & 1
            const char* At = lds + (step >> 1) * TILE;
            f16x8 a[TM];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const f16x8*>(At + ((a_rd + i * 2048) ^ ((s & 1) << 6)));
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int i = 0; i < TM; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[s][j], a[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int j = 0; j < TN; j++) bq[s][j] = ldb(rsW, lo, (step + D) * kstep + j * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: (acc + bias) + residual -> f16 -> y (and a second destination) --------------------------------------------------------------
    f16x4 o[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const long m = m0 + wm0 + i * 16 + l16;
            const int n = n0 + wn0 + j * 16 + g * 4;
#pragma unroll
            for (int r = 0; r < 4; r++) o[i][j][r] = (f16)((acc[i][j][r] + (float)bv[j][r]) + (float)rv[i][j][r]);
            *reinterpret_cast<f16x4*>(p.y + m * p.ldy + n) = o[i][j];
            if (p.y2) *reinterpret_cast<f16x4*>(p.y2 + m * p.ldy2 + n) = o[i][j];
        }
    if constexpr (TN % 2 == 0) {
        if (p.rs_out) {      // the four 16-lane groups of a row hold different columns of the same slot: 2-step butterfly, one group stores (gemm_epilogue_fast's order)
            const int np = N >> 5;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int h = 0; h < TN / 2; h++) {
                    float sm = 0.f, q = 0.f;
#pragma unroll
                    for (int jj = 0; jj < 2; jj++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float f = (float)o[i][2 * h + jj][r];
                            sm += f;
                            q = fmaf(f, f, q);
                        }
                    sm += __shfl_xor(sm, 16, 64); q += __shfl_xor(q, 16, 64);
                    sm += __shfl_xor(sm, 32, 64); q += __shfl_xor(q, 32, 64);
                    if (g == 0) {
                        float* d = p.rs_out + ((long)(m0 + wm0 + i * 16 + l16) * np + ((n0 + wn0) >> 5) + h) * 2;
                        d[0] = sm;
                        d[1] = q;
                    }
                }
        }
    }
}

template <int BM, int WGM, int WGN, int TN, bool LN>
int launch(osg_ctx* ctx, LinParams& p) {
    constexpr int BN = 16 * TN * WGN;
    auto kern = linear_small_kernel<BM, WGM, WGN, TN, LN>;
    const int smem = (p.K >> 6) * BM * 128 + (LN ? 4 * p.K : 0);
    static int attr_set = 0;
    if (attr_set < smem) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = smem;
    }
    p.mt = p.M / BM;
    p.nt = p.N / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.mt * p.nt)), dim3(256), (size_t)smem, ctx->compute, p);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

// tile choice: the row block must fit 80 KB of LDS (two workgroups per CU without LayerNorm operands); wider column tiles where the grid would
// otherwise be several rounds of workgroups.  cfg: 0 = 64 x 64, 1 = 64 x 128, 2 = 32 x 128, 3 = 16 x 128, 4 = 32 x 64, 5 = 16 x 64, -1 = not taken
inline int choose(int M, int N, int K, bool ln) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 320 || K > 2560) return -1;
    static const int forced = getenv("OSG_LINSMALL_CFG") ? atoi(getenv("OSG_LINSMALL_CFG")) : -1;
    auto ok = [&](int cfg) {
        const int bm = cfg <= 1 ? 64 : (cfg == 2 || cfg == 4) ? 32 : 16, bn = (cfg == 0 || cfg >= 4) ? 64 : 128;
        return M % bm == 0 && N % bn == 0 && (long)bm * K * 2 <= 80 * 1024;
    };
    if (forced >= 0) return ok(forced) ? forced : -1;
    if (K <= 640) {
        const long tiles64 = (long)(M / 64) * (N / 64);
        if (M % 64 == 0 && N % 128 == 0 && tiles64 > (ln ? 512 : 768)) return 1;
        if (ok(0)) return 0;
        if (ok(1)) return 1;
    }
    if (K <= 1280 && ok(2)) return 2;
    if (K <= 1280 && ok(4)) return 4;
    if (ok(3)) return 3;
    if (ok(5)) return 5;
    return -1;
}

}   // namespace osg_ls

extern "C" {

int osg_linear_small_supported(int M, int N, int K, int layer_norm) { return osg_ls::choose(M, N, K, layer_norm != 0) >= 0; }
// the same question for a launch that must also emit row statistics (its tile must hold whole 32-column slots)
int osg_linear_small_rowstats_supported(int M, int N, int K) { const int c = osg_ls::choose(M, N, K, false); return c >= 0 && c <= 3; }

int osg_linear_small(osg_ctx* ctx, const void* x, long ldx, const void* w_kn8, const void* bias, const void* residual, long ldr, const void* gamma,
                     const void* beta, float eps, void* y, long ldy, void* y2, long ldy2, int M, int N, int K, float* rowstats) {
    const bool ln = gamma != nullptr;
    const int cfg = osg_ls::choose(M, N, K, ln);
    if (cfg < 0) OSG_FAIL(ctx, "osg_linear_small: shape not taken (see osg_linear_small_supported)");
    if (ln && !beta) OSG_FAIL(ctx, "osg_linear_small: LayerNorm needs gamma and beta");
    if ((ldx && ldx < K) || (ldx % 8) || (ldy % 4) || (ldy2 % 4) || (ldr % 4)) OSG_FAIL(ctx, "osg_linear_small: row pitches must be multiples of 8 (x) / 4 elements");
    osg_ls::LinParams p;
    p.x = (const f16*)x; p.ldx = ldx ? ldx : K;
    p.w = (const f16*)w_kn8; p.bias = (const f16*)bias; p.res = (const f16*)residual; p.ldr = ldr ? ldr : N;
    p.gamma = (const f16*)gamma; p.beta = (const f16*)beta; p.eps = eps;
    p.y = (f16*)y; p.ldy = ldy ? ldy : N; p.y2 = (f16*)y2; p.ldy2 = ldy2;
    p.rs_out = rowstats;
    if (rowstats && cfg > 3) OSG_FAIL(ctx, "osg_linear_small: this shape's tile cannot emit row statistics (osg_linear_small_rowstats_supported)");
    p.M = M; p.N = N; p.K = K;
    using namespace osg_ls;
    if (ln) {
        switch (cfg) {
            case 0: return launch<64, 2, 2, 2, true>(ctx, p);
            case 1: return launch<64, 2, 2, 4, true>(ctx, p);
            case 2: return launch<32, 1, 4, 2, true>(ctx, p);
            case 3: return launch<16, 1, 4, 2, true>(ctx, p);
            case 4: return launch<32, 1, 4, 1, true>(ctx, p);
            default: return launch<16, 1, 4, 1, true>(ctx, p);
        }
    }
    switch (cfg) {
        case 0: return launch<64, 2, 2, 2, false>(ctx, p);
        case 1: return launch<64, 2, 2, 4, false>(ctx, p);
        case 2: return launch<32, 1, 4, 2, false>(ctx, p);
        case 3: return launch<16, 1, 4, 2, false>(ctx, p);
        case 4: return launch<32, 1, 4, 1, false>(ctx, p);
        default: return launch<16, 1, 4, 1, false>(ctx, p);
    }
}

}   // extern "C"
