// libosgpu: the halo-reuse 3x3 convolution with the weights resident as uint8 CODES (W8A16, round 6): the WQ = 1 instantiations of conv3x3_kernel
// (osg_conv3x3_kernel.h).  The [BN][64] weight tile streams HBM -> L2 -> LDS as codes -- half the bytes of the operand the 8 x 8 / 16 x 16 levels are bound by -- and
// becomes halves between LDS and the MFMA (osg_gemm_common.h w8_frag: exact integers q - zp, the scale applied once to the f32 accumulators).  Tile choice, split-K,
// fold, statistics sinks, output views: those of the f16 kernel (osg_conv3x3.hip drives both).  Reference: get_tensor_data dequantises a uint8 weight when it is
// loaded, src/onnxstream.cpp:2887-2891 -> Model::dequantize :3353; the convolution itself :1292-1534.
#include "osg_conv3x3_kernel.h"

namespace {

template <int W_>
int launch3_w8(osg_ctx* ctx, GemmParams& p, int bn) {
    if (bn == 80) return launch3<W_, 80, 4, 1, 0, 4, 1>(ctx, p);
    if (bn == 160) {
        // (64-pixel rows with 160 columns: one register short with the code registers of the three-step B pipeline, and 128 tiles for 256 CUs at the only width it
        // divides, 320 -- the 80-column tile runs instead)
        if constexpr (W_ == 64) return launch3<W_, 80, 4, 1, 0, 4, 1>(ctx, p);
        else return launch3<W_, 160, 2, 2, 0, 4, 1>(ctx, p);
    }
    return launch3<W_, 128, 2, 2, 0, 4, 1>(ctx, p);
}

}  // namespace

int osg_conv3x3_w8_tile(osg_ctx* ctx, GemmParams& p, int bn) {
    if (p.W == 64) return launch3_w8<64>(ctx, p, bn);
    if (p.W == 32) return launch3_w8<32>(ctx, p, bn);
    if (p.W == 16) return launch3_w8<16>(ctx, p, bn);
    return launch3_w8<8>(ctx, p, bn);
}
