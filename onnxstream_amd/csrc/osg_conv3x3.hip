// libosgpu: halo-reuse 3x3 / stride 1 / pad 1 convolution (the resnet convolutions: 55 % of an SD UNet pass).
//
// Measured on MI355X the implicit-GEMM kernel of osg_gemm.hip is bound by the per-CU L2->LDS path (~23 B/clk/CU), not by
// the MFMA pipe, and a 3x3 convolution run as a GEMM fetches every input pixel NINE times (once per filter tap).  Here a
// workgroup owns 128 consecutive output pixels (TH full rows of the image) x BN output channels and walks K as
// (64-channel slab) x (9 taps):
//   * the input PATCH of the slab -- (TH+2) x (W+2) pixels x 64 channels, zero halo supplied by the buffer descriptor's
//     bounds check -- is DMA'd into LDS ONCE and read by all 9 taps (A traffic / ~5);
//   * per (slab, tap) only the [BN][64] weight tile streams in: 4-deep LDS ring, counted vmcnt, ONE barrier per tap;
//   * the patch of the NEXT slab arrives piecewise during taps 0..4 of the current slab (double-buffered);
//   * both images are lane-linear 128-byte rows with the 16-byte chunk index XOR-swizzled by (row & 7) on the SOURCE address,
//     so every fragment read is a conflict-free ds_read_b128 (a fragment's 16 pixels are consecutive patch pixels);
//   * BN in {80, 128, 160}: 320 / 640 / 1280 output channels tile without padding; wave layouts 4x1 / 2x2 / 2x2.
// Arithmetic, accumulation order class and epilogue are those of the GEMM kernel (f16 operands, f32 accumulate on
// v_mfma_f32_16x16x32_f16, one RNE rounding): XnnPack::convolution, reference src/onnxstream.cpp:1292-1534.
#include "osg_conv3x3_kernel.h"

namespace {

template <int W_>
int launch3_bn(osg_ctx* ctx, GemmParams& p, int bn, int nl) {
    if (nl == 8) {   // 8 loader waves (768 threads): a measured candidate only (osg_tune.h) -- the same arithmetic, twice the DMA issue slots
        // (where the wider stages leave the LDS ring too short -- W = 8 with BN = 160 -- the 4-loader kernel runs instead)
        if (bn == 80) { if constexpr (Geo<W_, 80, 8>::OK) return launch3<W_, 80, 4, 1, 0, 8>(ctx, p); }
        else if (bn == 160) { if constexpr (Geo<W_, 160, 8>::OK) return launch3<W_, 160, 2, 2, 0, 8>(ctx, p); }
        else { if constexpr (Geo<W_, 128, 8>::OK) return launch3<W_, 128, 2, 2, 0, 8>(ctx, p); }
    }
    if (bn == 80) return launch3<W_, 80, 4, 1>(ctx, p);
    if (bn == 160) return launch3<W_, 160, 2, 2>(ctx, p);
    return launch3<W_, 128, 2, 2>(ctx, p);
}

}  // namespace

int osg_conv3x3_supported(int N, int H, int W, int Cin, int Cout) {
    if (!(W == 64 || W == 32 || W == 16 || W == 8)) return 0;
    const int TI = W == 8 ? 2 : 1, TH = 128 / (W * TI);
    if (H % TH || (W == 8 && H != 8)) return 0;
    if (Cin % 64 || Cout % 4) return 0;
    if ((double)N * H * W * Cin * 2.0 >= 2147483648.0 || (double)Cout * 9.0 * Cin * 2.0 >= 2147483648.0) return 0;
    return 1;
}

// Shape gate + tile/split choice.  Model (cycles, calibrated like choose_v2 in osg_gemm.hip): a (slab, tap) unit costs
// max(MFMA, bytes / 23 B/clk) with bytes = the weight tile + 1/9 of the patch; whole rounds of tiles over the CUs.
// shape gate: -1 when the halo-reuse kernel does not take the shape; fills the buffer extents otherwise
int osg_conv3x3_prepare(osg_ctx* ctx, GemmParams& p) {
    static const bool off = getenv("OSG_CONV3X3_OFF") != nullptr;
    if (off) return -1;
    const int W = p.W, H = p.H;
    if (!(W == 64 || W == 32 || W == 16 || W == 8)) return -1;
    if (p.Wo != W || p.Ho != H || p.sh != 1 || p.sw != 1 || p.pt != 1 || p.pl != 1 || p.KW != 3 || p.K != 9 * p.Cin) return -1;
    if (p.Cin % 64 || p.N % 4) return -1;
    const int TI = W == 8 ? 2 : 1, TH = 128 / (W * TI);
    if (H % TH) return -1;
    if (W == 8 && H != 8) return -1;
    if (p.a_bytes_l >= 2147483648L || (double)p.N * p.K * 2.0 >= 2147483648.0) return -1;
    if ((((uintptr_t)p.A | (uintptr_t)p.Bt) & 15) != 0) return -1;
    p.a_bytes = (unsigned)p.a_bytes_l;
    p.b_bytes = (unsigned)((long)p.N * p.K * (p.w8 ? 1 : 2));
    return 0;
}

// every legal (BN, splits) with its modelled cost in cycles, cheapest first
std::vector<std::pair<double, std::pair<int, int>>> osg_conv3x3_rank(const osg_ctx* ctx, const GemmParams& p) {
    const int W = p.W;
    const int TI = W == 8 ? 2 : 1, TH = 128 / (W * TI);
    const double cus = ctx->num_cu;
    const int slabs = p.Cin / 64;
    const int mt = (p.M + 127) / 128;
    const int pp = TI * (TH + 2) * (W == 8 ? 16 : W + 2);
    std::vector<std::pair<double, std::pair<int, int>>> out;
    static const int bns[3] = {80, 128, 160};
    for (int bn : bns) {
        if (bn != 128 && p.N % bn) continue;
        const double tiles = (double)mt * ((p.N + bn - 1) / bn);
        const double mfma = 128.0 * bn * 128.0 / 4069.0;
        const double tload = (bn * (p.w8 ? 64.0 : 128.0) + pp * 128.0 / 9.0) / 23.0;
        for (int s = 1; s <= (ctx->autotune ? 12 : 8); s++) {
            if (s > 1 && slabs / s < (ctx->autotune ? 1 : 2)) break;   // measured choice: let finer splits compete too
            const int sl = (slabs + s - 1) / s;
            if (s > 1 && sl * (s - 1) >= slabs) continue;
            const double blocks = tiles * s;
            const double rounds = std::ceil(blocks / cus);
            double cost = rounds * (sl * 9.0 * (std::max(mfma, tload) + 250.0) + 6000.0);
            if (s > 1) cost += 9000.0 + (double)p.M * p.N * s * 4.0 / 2000.0;
            out.push_back({cost, {bn, s}});
            // (measured candidates only) the same split finished inside the kernel by the last arriver of each tile: splits + 1000
            if (ctx->autotune && s >= 2 && s <= 4 && osg_mm::splitk_fold_mode() != 0) out.push_back({cost * 1.0005, {bn, s + 1000}});
        }
    }
    std::stable_sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    return out;
}

// launch one configuration (reduce kernel included); p must have passed osg_conv3x3_prepare
int osg_conv3x3_launch(osg_ctx* ctx, GemmParams p, int bn, int s, int nl, int fold) {
    const int slabs = p.Cin / 64;
    if (s < 1) s = 1;
    const int sl = (slabs + s - 1) / s;
    p.splits = (slabs + sl - 1) / sl;
    p.k_per_split = sl * 64;
    p.tickets = nullptr;
    p.fold_acc = 0;
    if (p.splits > 1) {
        size_t need = (size_t)p.splits * p.M * p.N * sizeof(float);
        if (fold) need = std::max(need, osg_mm::splitk_fold_route(ctx, p, (long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn), 128, bn));
        if (osg_ensure_workspace(ctx, need)) return 1;
        p.partial = (float*)ctx->ws;
    }
    p.n_major = (double)p.N * p.K * 2.0 > (double)p.a_bytes_l;
    int rc;
    if (p.w8) rc = osg_conv3x3_w8_tile(ctx, p, bn);   // (uint8 weight codes: osg_conv3x3_w8.hip, 4 loader waves)
    else if (p.W == 64) rc = launch3_bn<64>(ctx, p, bn, nl);
    else if (p.W == 32) rc = launch3_bn<32>(ctx, p, bn, nl);
    else if (p.W == 16) rc = launch3_bn<16>(ctx, p, bn, nl);
    else rc = launch3_bn<8>(ctx, p, bn, nl);
    if (rc) return rc;
    if (p.splits > 1 && !p.fold_acc) return launch_splitk_reduce(ctx, p, 1);
    return 0;
}

int osg_conv3x3_run(osg_ctx* ctx, GemmParams& p) {
    if (osg_conv3x3_prepare(ctx, p)) return -1;
    auto ranked = osg_conv3x3_rank(ctx, p);
    int best_bn = 128, best_s = 1;
    if (!ranked.empty()) { best_bn = ranked[0].second.first; best_s = ranked[0].second.second % 1000; }
    if (const char* e = getenv("OSG_CONV3X3_BN")) best_bn = atoi(e);
    if (const char* e = getenv("OSG_CONV3X3_SPLITS")) best_s = atoi(e);
    int nl = 4;
    if (const char* e = getenv("OSG_CONV3X3_NL")) nl = atoi(e) == 8 ? 8 : 4;
    const int fold = getenv("OSG_CONV3X3_FOLD") ? atoi(getenv("OSG_CONV3X3_FOLD")) : 0;   // (tests / probes)
    return osg_conv3x3_launch(ctx, p, best_bn, best_s, nl, fold);
}
