// libosgpu: W8A16 contractions -- uint8 weights stay RESIDENT AS uint8 (half the HBM/L2 bytes of the dominant operand) and are
// dequantised on the fly into the f16 LDS tile the MFMAs read.
//
// Reference semantics (src/onnxstream.cpp:2887-2891 -> Model::dequantize :3353): a uint8 weight with per-tensor (scale, zero_point)
// becomes  w = f16( (float)((int)q - zp) * scale )  when it is loaded, and the f16 GEMM/conv then runs as usual.  The kernel below
// computes exactly that value for every weight element -- only later and on chip -- so its results are those of the f16 kernel on the
// dequantised weights (same MFMA order).
//
// Structure = the wave-specialised form of gemm2_kernel (osg_gemm.hip): 512 threads, 4 MATH waves (ds_read + MFMA + epilogue) and
// 4 LOADER waves.  Per 64-deep k-tile a loader wave
//   * DMAs its share of the f16 A tile (activations / implicit-GEMM pixel gather) straight into LDS (`buffer_load ... lds`), and
//   * fetches its share of the [BN][64] uint8 weight tile with ordinary 16-byte buffer loads into a 3-deep VGPR ring (the k loop is
//     unrolled by the ring depth so every register index is static), converts 16 codes -> 16 halves with v_cvt_f32_ubyteN / sub / mul /
//     cvt_f16 and writes two swizzled 16-byte chunks with ds_write_b128.
// Both kinds of load share the vmcnt queue, so ONE counted wait covers tile kt's A image and its weight registers.
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

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 dequant8(int lo, int hi, float zpf, float scale) {
    f16x8 r;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        r[b] = (f16)(((float)((lo >> (8 * b)) & 0xff) - zpf) * scale);
        r[4 + b] = (f16)(((float)((hi >> (8 * b)) & 0xff) - zpf) * scale);
    }
    return r;
}

template <int BN, bool CONV>
__device__ __forceinline__ void gemm_w8_body(const GemmParams& p) {
    constexpr int BM = 128, NST = 4, D = NST - 1, ROWB = 128;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LD = BM / 32;            // 1-KiB DMA wave-loads per loader wave per k-tile (A)
    constexpr int B_LDW = BN / 64;           // 16-byte register loads per loader lane per k-tile (B: 16 rows x 64 codes per wave-load)
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int INFLIGHT = (D - 1) * (A_LD + B_LDW);
    constexpr unsigned OOB = 0x80000000u;

    extern __shared__ __attribute__((aligned(16))) char smem5[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;

    int L;
    {
        const int total = gridDim.x, bid = blockIdx.x, x = bid & 7, i = bid >> 3, q = total >> 3, r = total & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    int m_tile, n_tile, zs;
    if (p.n_major) {
        n_tile = L / (p.splits * p.mt); L -= n_tile * p.splits * p.mt;
        zs = L / p.mt; m_tile = L - zs * p.mt;
    } else {
        m_tile = L / (p.splits * p.nt); L -= m_tile * p.splits * p.nt;
        zs = L / p.nt; n_tile = L - zs * p.nt;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int kbeg = zs * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nkt = (kend - kbeg) >> 6;

    if (loader) {
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bt, 0, p.b_bytes, 0x00020000);
        const float zpf = (float)p.w_zp, scale = p.w_scale;
        // ---- A: DMA addressing (as gemm2_kernel) --------------------------------------------------------------------------
        const int rsub = lane >> 3;
        const int gch = (lane & 7) ^ rsub;
        int a_base[A_LD], a_hi0[A_LD], a_wi0[A_LD];
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
        // ---- B: uint8 rows; wave-load g = j*4 + wave covers rows g*16 + (lane>>2), 16 codes at column (lane&3)*16 -----------------
        unsigned b_off[B_LDW];
        int b_lds[B_LDW][2];
#pragma unroll
        for (int j = 0; j < B_LDW; j++) {
            const int row = (j * 4 + wave) * 16 + (lane >> 2);
            const int c16 = lane & 3;
            const int n = n0 + row;
            b_off[j] = n < p.N ? (unsigned)((long)n * p.K + c16 * 16) : OOB;
            b_lds[j][0] = A_BYTES + row * ROWB + (((2 * c16) ^ (row & 7)) << 4);
            b_lds[j][1] = A_BYTES + row * ROWB + (((2 * c16 + 1) ^ (row & 7)) << 4);
        }
        int ik = kbeg, i_c0 = 0, i_kh = 0, i_kw = 0;
        if (CONV) {
            const int cell = kbeg / p.Cin;
            i_c0 = kbeg - cell * p.Cin;
            i_kh = cell / p.KW;
            i_kw = cell - i_kh * p.KW;
        }
        v4i breg[D][B_LDW];
        auto issue_tile = [&](int stage, v4i (&br)[B_LDW]) {      // next tile in sequence: A by DMA into `stage`, B into registers
            char* As = smem5 + stage * STAGE;
            const bool live = ik < kend;
            const unsigned kill = live ? 0u : OOB;
            if (CONV) {
                const int tap_off = ((i_kh * p.W + i_kw) * p.Cin + i_c0) * 2;
#pragma unroll
                for (int j = 0; j < A_LD; j++) {
                    const int hi = a_hi0[j] + i_kh, wi = a_wi0[j] + i_kw;
                    const bool ok = live && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    const unsigned off = ok ? (unsigned)(a_base[j] + tap_off) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(As + (j * 4 + wave) * 1024), 16, off, 0, 0, 0);
                }
                i_c0 += 64;
                if (i_c0 >= p.Cin) { i_c0 = 0; if (++i_kw == p.KW) { i_kw = 0; ++i_kh; } }
            } else {
#pragma unroll
                for (int j = 0; j < A_LD; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(As + (j * 4 + wave) * 1024), 16, (unsigned)a_base[j] | kill, ik * 2, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < B_LDW; j++) br[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, b_off[j] | kill, ik, 0);
            ik += 64;
        };
        auto commit_b = [&](v4i (&br)[B_LDW], int stage) {         // dequantise + write the weight tile's share into `stage`
            char* St = smem5 + stage * STAGE;
#pragma unroll
            for (int j = 0; j < B_LDW; j++) {
                *reinterpret_cast<f16x8*>(St + b_lds[j][0]) = dequant8(br[j][0], br[j][1], zpf, scale);
                *reinterpret_cast<f16x8*>(St + b_lds[j][1]) = dequant8(br[j][2], br[j][3], zpf, scale);
            }
        };
        static_for<0, D>([&](auto d) { issue_tile(decltype(d)::value, breg[decltype(d)::value]); });
        for (int kt0 = 0; kt0 < nkt; kt0 += D) {
            static_for<0, D>([&](auto ph) {
                constexpr int s = decltype(ph)::value;
                const int kt = kt0 + s;
                if (kt < nkt) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // tile kt: my A DMAs landed, my weight registers ready
                    commit_b(breg[s], kt & (NST - 1));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                                      // tile kt complete in LDS; tile kt-1's stage is free
                    issue_tile((kt + D) & (NST - 1), breg[s]);
                }
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- math waves --------------------------------------------------------------------------------------------------------
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int fsw = (((lane >> 4) ^ (frow & 7)) << 4);
    const int a_rd = (wm0 + frow) * ROWB + fsw;
    const int b_rd = A_BYTES + (wn0 + frow) * ROWB + fsw;
    for (int kt = 0; kt < nkt; kt++) {
        __builtin_amdgcn_s_barrier();
        const char* St = smem5 + (kt & (NST - 1)) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            f16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const f16x8*>(St + ((a_rd + i * 16 * ROWB) ^ (ks << 6)));
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = *reinterpret_cast<const f16x8*>(St + ((b_rd + j * 16 * ROWB) ^ (ks << 6)));
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    gemm_epilogue<TM, TN>(p, acc, m0, n0, wm0, wn0, lane, 0, zs);
}

template <int BN, bool CONV>
__global__ __launch_bounds__(512) void gemm_w8_kernel(GemmParams p) {
    gemm_w8_body<BN, CONV>(p);
}

template <int BN, bool CONV>
int launch_w8(osg_ctx* ctx, GemmParams& p) {
    constexpr size_t smem = (size_t)4 * (128 + BN) * 128;
    auto kern = gemm_w8_kernel<BN, CONV>;
    static unsigned long long attr_mask = 0;   // (per device: hipFuncSetAttribute is, and a process may hold several)
    if (osg_first_on_device(attr_mask)) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    p.mt = (p.M + 127) / 128;
    p.nt = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.mt * p.nt * p.splits)), dim3(512), smem, ctx->compute, p);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

template <bool CONV>
int run_w8(osg_ctx* ctx, GemmParams p) {
    const int bn = p.N > 64 ? 128 : 64;
    // split-K by the same model as the f16 kernel (tile bytes are 3/4: 16 KiB of A + 8 KiB of codes per 128x128x64 k-tile)
    const double cus = ctx->num_cu;
    const int kt = p.K / 64;
    const double tiles = (double)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
    int best_s = 1;
    double best = 1e300;
    for (int s = 1; s <= 16; s++) {
        if (s > 1 && kt / s < 8) break;
        const int kts = (kt + s - 1) / s;
        if (s > 1 && kts * (s - 1) >= kt) continue;
        const double rounds = std::ceil(tiles * s / cus);
        const double tk = std::max(128.0 * bn * 128.0 / 4069.0, (128 * 128.0 + bn * 64.0) / 23.0) + 300.0;
        double cost = rounds * (kts * tk + 4000.0);
        if (s > 1) cost += 9000.0 + (double)p.M * p.N * s * 4.0 / 2000.0;
        if (cost < best) { best = cost; best_s = s; }
    }
    const int kt_per = (kt + best_s - 1) / best_s;
    p.splits = (kt + kt_per - 1) / kt_per;
    p.k_per_split = kt_per * 64;
    p.tickets = nullptr;
    if (p.splits > 1) {
        if (osg_ensure_workspace(ctx, (size_t)p.splits * p.M * p.N * sizeof(float))) return 1;
        p.partial = (float*)ctx->ws;
    }
    const double a_unique = CONV ? (double)p.a_bytes_l : (double)p.M * p.K * 2.0;
    p.n_major = (double)p.N * p.K > a_unique;
    int rc = bn == 128 ? launch_w8<128, CONV>(ctx, p) : launch_w8<64, CONV>(ctx, p);
    if (rc) return rc;
    if (p.splits > 1) return launch_splitk_reduce(ctx, p, 1);
    return 0;
}

}  // namespace

extern "C" {

int osg_gemm_w8(osg_ctx* ctx, const void* A, const void* Bq_nk, float w_scale, int w_zero_point, const void* bias, osg_dtype bias_dtype,
                const void* residual, void* C, int M, int N, int K, osg_act act) {
    if (M <= 0 || N <= 0 || K <= 0) OSG_FAIL(ctx, "osg_gemm_w8: invalid shape of inputs");
    if (K % 64 || ((uintptr_t)A & 15) || ((uintptr_t)Bq_nk & 15)) OSG_FAIL(ctx, "osg_gemm_w8: K must be a multiple of 64 and the operands 16-byte aligned");
    if (bias && bias_dtype != OSG_F16 && bias_dtype != OSG_F32) OSG_FAIL(ctx, "osg_gemm_w8: invalid bias dtype");
    if (act == OSG_ACT_GEGLU) OSG_FAIL(ctx, "osg_gemm_w8: the GEGLU epilogue needs interleaved f16 weights");
    if ((double)M * K * 2.0 >= 2147483648.0 || (double)N * K >= 2147483648.0) OSG_FAIL(ctx, "osg_gemm_w8: operand larger than 2 GiB");
    GemmParams p{};
    p.A = (const f16*)A; p.Bt = (const f16*)Bq_nk; p.C = (f16*)C; p.bias = bias; p.residual = (const f16*)residual;
    p.M = M; p.N = N; p.K = K; p.lda = K;
    p.bias_f32 = bias_dtype == OSG_F32; p.act = act;
    p.w_scale = w_scale; p.w_zp = w_zero_point;
    p.a_bytes = (unsigned)(((long)(M - 1) * K + K) * 2);
    p.b_bytes = (unsigned)((long)N * K);
    return run_w8<false>(ctx, p);
}

int osg_conv2d_nhwc_w8(osg_ctx* ctx, const void* x, const void* wq_ohwi, float w_scale, int w_zero_point, const void* bias,
                       osg_dtype bias_dtype, const void* image_bias, long image_bias_ld, const void* residual, void* y, int N, int H, int W,
                       int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl, int pb, int pr, osg_act act) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0) OSG_FAIL(ctx, "osg_conv2d_nhwc_w8: invalid argument");
    if (Cin % 64 || ((uintptr_t)x & 15) || ((uintptr_t)wq_ohwi & 15)) OSG_FAIL(ctx, "osg_conv2d_nhwc_w8: Cin must be a multiple of 64 and the operands 16-byte aligned");
    if (bias && bias_dtype != OSG_F16 && bias_dtype != OSG_F32) OSG_FAIL(ctx, "osg_conv2d_nhwc_w8: invalid bias dtype");
    const int Ho = (H + pt + pb - KH) / sh + 1, Wo = (W + pl + pr - KW) / sw + 1;
    if (Ho <= 0 || Wo <= 0) OSG_FAIL(ctx, "osg_conv2d_nhwc_w8: empty output");
    GemmParams p{};
    p.A = (const f16*)x; p.Bt = (const f16*)wq_ohwi; p.C = (f16*)y; p.bias = bias; p.residual = (const f16*)residual;
    p.M = N * Ho * Wo; p.N = Cout; p.K = KH * KW * Cin; p.lda = 0;
    p.bias_f32 = bias_dtype == OSG_F32; p.act = act;
    p.w_scale = w_scale; p.w_zp = w_zero_point;
    p.a_bytes_l = (long)N * H * W * Cin * 2;
    if (p.a_bytes_l >= 2147483648L || (double)Cout * p.K >= 2147483648.0) OSG_FAIL(ctx, "osg_conv2d_nhwc_w8: operand larger than 2 GiB");
    p.a_bytes = (unsigned)p.a_bytes_l;
    p.b_bytes = (unsigned)((long)Cout * p.K);
    p.rowbias = (const f16*)image_bias; p.rb_rows = Ho * Wo; p.rb_ld = image_bias_ld;
    p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.KW = KW; p.sh = sh; p.sw = sw; p.pt = pt; p.pl = pl;
    return run_w8<true>(ctx, p);
}

}  // extern "C"
