// libosgpu: W8A16 contractions with the weight CODES resident (round 6 rewrite): the WQ = 1 instantiations of the direct-to-LDS contraction kernel (gemm2_kernel,
// osg_gemm2.h) for Linear / MatMul / Gemm and the 1 x 1 convolutions; the implicit-GEMM convolutions are in osg_gemm_w8_conv.hip, the halo-reuse 3 x 3
// convolution in osg_conv3x3_w8.hip.
//
// Reference semantics (src/onnxstream.cpp:2887-2891 -> Model::dequantize :3353): a uint8 weight with (scale, zero_point) becomes w = f16((float)((int)q - zp) * scale)
// when it is LOADED and the f16 GEMM / convolution then runs as usual.  Here the codes stay uint8 in HBM (half the footprint), stream through L2 and the LDS ring as
// codes (half the bytes of the weight operand on every hop -- the operand the 8 x 8 / 16 x 16 levels of the UNet are bound by, and half the bytes through the LDS
// port the k loop is bound by), and become halves between the LDS tile and the MFMA: the integer q - zp exactly (osg_gemm_common.h w8_frag), the scale applied once to
// the f32 accumulators.  Tiles, rings, split-K, the in-kernel fold, GEGLU, output views, statistics sinks: those of the f16 kernel -- osg_gemm.hip drives both,
// W8 shapes have rows of their own in the tune table (flag 1024 of the key).  Rounds 3-5 had a lean 512-thread kernel here that dequantised in its loader waves
// (register staging, ds_write): 1.6 x the f16 plan on the SD 1.5 pass.
#include "osg_gemm2.h"

namespace osg_mm {

#define OSG_W8(BM_, BN_, NST_, WGN_) return launch_v2<BM_, BN_, NST_, false, 0, 0, 0, 5, 1, WGN_, 1>(ctx, p, batch)
static int launch_v2_w8_plain(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst) {
    switch (tile) {
    case 0: if (nst == 2) OSG_W8(128, 128, 2, 2); if (nst == 4) OSG_W8(128, 128, 4, 2); break;
    case 1: if (nst == 2) OSG_W8(128, 64, 2, 2); if (nst == 4) OSG_W8(128, 64, 4, 2); if (nst == 6) OSG_W8(128, 64, 6, 2); break;
    case 2: if (nst == 2) OSG_W8(64, 64, 2, 2); if (nst == 4) OSG_W8(64, 64, 4, 2); if (nst == 8) OSG_W8(64, 64, 8, 2); break;
    case 3: if (nst == 2) OSG_W8(64, 128, 2, 2); if (nst == 4) OSG_W8(64, 128, 4, 2); break;
    case 4: if (nst == 2) OSG_W8(128, 160, 2, 1); if (nst == 4) OSG_W8(128, 160, 4, 1); break;
    case 5: if (nst == 2) OSG_W8(128, 80, 2, 1); if (nst == 4) OSG_W8(128, 80, 4, 1); break;
    case 6: if (nst == 2) OSG_W8(64, 80, 2, 1); if (nst == 4) OSG_W8(64, 80, 4, 1); if (nst == 6) OSG_W8(64, 80, 6, 1); break;
    case 7: if (nst == 2) OSG_W8(64, 160, 2, 2); if (nst == 4) OSG_W8(64, 160, 4, 2); break;
    }
    return -2;
}
#undef OSG_W8

int launch_v2_w8_conv(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst);   // osg_gemm_w8_conv.hip

int launch_v2_w8(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst, bool conv) {
    if (p.ln_c1 || p.rs_out) return -2;                                  // (LayerNorm folds gamma into f16 weights; row statistics: the f16 kernels)
    if (p.act == OSG_ACT_GEGLU && tile >= 5) return -2;                  // (an even number of 16-column blocks per wave)
    return conv ? launch_v2_w8_conv(ctx, p, batch, tile, nst) : launch_v2_w8_plain(ctx, p, batch, tile, nst);
}

bool w8_tile_has(int tile, int nst, bool conv) {
    if (tile < 0 || tile > 7) return false;
    if (conv) return tile <= 2 ? (nst == 2 || nst == 4) : (tile <= 6 && tile != 3 && nst == 4);
    if (nst == 2 || nst == 4) return true;
    return (nst == 6 && (tile == 1 || tile == 6)) || (nst == 8 && tile == 2);
}

}  // namespace osg_mm
