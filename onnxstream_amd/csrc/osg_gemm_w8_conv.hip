// libosgpu: the implicit-GEMM convolutions with the weight codes resident (W8A16; see osg_gemm_w8.hip): WQ = 1, CONV instantiations of gemm2_kernel -- the
// stride-2 downsampling convolutions and every 3 x 3 shape the halo-reuse kernel does not take.
#include "osg_gemm2.h"

namespace osg_mm {

#define OSG_W8C(BM_, BN_, NST_, WGN_) return launch_v2<BM_, BN_, NST_, true, 0, 0, 0, 5, 1, WGN_, 1>(ctx, p, batch)
int launch_v2_w8_conv(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst) {
    switch (tile) {
    case 0: if (nst == 2) OSG_W8C(128, 128, 2, 2); if (nst == 4) OSG_W8C(128, 128, 4, 2); break;
    case 1: if (nst == 2) OSG_W8C(128, 64, 2, 2); if (nst == 4) OSG_W8C(128, 64, 4, 2); break;
    case 2: if (nst == 2) OSG_W8C(64, 64, 2, 2); if (nst == 4) OSG_W8C(64, 64, 4, 2); break;
    case 4: if (nst == 4) OSG_W8C(128, 160, 4, 1); break;
    case 5: if (nst == 4) OSG_W8C(128, 80, 4, 1); break;
    case 6: if (nst == 4) OSG_W8C(64, 80, 4, 1); break;
    }
    return -2;
}
#undef OSG_W8C

}  // namespace osg_mm
