// libosgpu: the 160- and 80-column tiles of the direct-to-LDS contraction kernel (round 6; kernel and launcher: osg_gemm2.h, choice: osg_gemm.hip).
//
// Why: every output width of the SD / SDXL UNets is a multiple of 320 (320 ... 10 240, 960 / 1 920 / 3 840 for the merged Q|K|V projections), none but 640 x 2^n
// a multiple of 128 -- and a batch-2 layer is about ONE tile per CU, so how the tile grid divides the 256 CUs decides a launch more than the k loop does:
// the GEGLU projection of the 16 x 16 level (512 x 10 240 x 1 280) is 4 x 80 = 320 tiles of 128 x 128 (two rounds on 64 CUs, one on the rest) but 4 x 64 = 256
// tiles of 128 x 160; the one of the 32 x 32 level (2 048 x 5 120 x 640) is 1 280 tiles of 128 x 64 on 768 resident slots but 512 of 128 x 160 on 512; a
// 2 048 x 640 output is 320 tiles of 64 x 64 but 256 of 64 x 80.  The wider tile also fetches fewer bytes per MFMA (128 x 160: 71 FLOP per byte, 128 x 64: 43).
//   tile 4: 128 x 160, waves 4 x 1 (32 rows x 160 columns each: 10 column blocks, even -- the GEGLU pairing works)      rings 2 / 4, LayerNorm-folded forms, CONV (4)
//   tile 5: 128 x  80, waves 4 x 1 (the halo convolution's wave tile, 32 x 80)                                           rings 2 / 4, CONV (4)
//   tile 6:  64 x  80, waves 4 x 1 (16 x 80)                                                                             rings 2 / 4 / 6, CONV (4)
//   tile 7:  64 x 160, waves 2 x 2 (32 x 80)                                                                             rings 2 / 4, LayerNorm-folded forms
// Measured candidates only (osg_tune.h): the cost-model plans of the parity tests never take them.  Arithmetic: the same MFMA sequence per output element as every
// other tile (k ascending in steps of 32, f32 accumulate, one rounding) -- a tile choice changes no bits unless it changes the split of K.
#include "osg_gemm2.h"

namespace osg_mm {

// returns -2 when the (tile, ring, form) asked for has no instantiation (the caller then takes one of the round-2 tiles)
int launch_v2_wide(osg_ctx* ctx, GemmParams& p, int batch, int tile, int nst, bool conv, int spec) {
    const bool ln2 = p.ln_c1 && p.rs_in, ln1 = p.ln_c1 && !p.rs_in;
    if (ln1) return -2;                                   // (row statistics beside the MFMAs: the round-2 tiles only)
    const int nch = ln2 ? (p.rs_np >> 1) : 0;
    const bool geglu = p.act == OSG_ACT_GEGLU;
    if (geglu && tile != 4) return -2;                    // (an even number of 16-column blocks per wave: the 4 x 1 form of the 160-column tile only)
    if (p.rs_out && tile != 4) return -2;                 // (row statistics per 32-column slot: a wave's columns must start at a multiple of 32 and be whole slots)
#define OSG_W(BM_, BN_, NST_, CONV_, LN_, NCH_, WGN_) return launch_v2<BM_, BN_, NST_, CONV_, 0, 0, LN_, NCH_, 1, WGN_>(ctx, p, batch)
#define OSG_W_LN2(BM_, BN_, NST_, WGN_)                                   \
    do {                                                                  \
        if (nch <= 5) OSG_W(BM_, BN_, NST_, false, 2, 5, WGN_);           \
        else if (nch <= 10) OSG_W(BM_, BN_, NST_, false, 2, 10, WGN_);    \
        else OSG_W(BM_, BN_, NST_, false, 2, 20, WGN_);                   \
    } while (0)
    if (conv) {
        if (ln2 || nst != 4) return -2;
        if (tile == 4) OSG_W(128, 160, 4, true, 0, 5, 1);
        if (tile == 5) OSG_W(128, 80, 4, true, 0, 5, 1);
        if (tile == 6) OSG_W(64, 80, 4, true, 0, 5, 1);
        return -2;
    }
    if (tile == 4 && spec && nst == 4) {   // four loader waves beside the four math waves (gemm2_kernel SPEC = 1): 144 KiB of ring, one workgroup per CU
        p.fold_acc = 0;
        if (ln2) {
            if (nch <= 5) return launch_v2<128, 160, 4, false, 0, 1, 2, 5, 1, 1>(ctx, p, batch);
            if (nch <= 10) return launch_v2<128, 160, 4, false, 0, 1, 2, 10, 1, 1>(ctx, p, batch);
            return launch_v2<128, 160, 4, false, 0, 1, 2, 20, 1, 1>(ctx, p, batch);
        }
        return launch_v2<128, 160, 4, false, 0, 1, 0, 5, 1, 1>(ctx, p, batch);
    }
    if (tile == 4) {
        if (ln2) { if (nst == 2) OSG_W_LN2(128, 160, 2, 1); if (nst == 4) OSG_W_LN2(128, 160, 4, 1); return -2; }
        if (nst == 2) OSG_W(128, 160, 2, false, 0, 5, 1);
        if (nst == 4) OSG_W(128, 160, 4, false, 0, 5, 1);
        return -2;
    }
    if (tile == 5) {
        if (ln2) return -2;
        if (nst == 2) OSG_W(128, 80, 2, false, 0, 5, 1);
        if (nst == 4) OSG_W(128, 80, 4, false, 0, 5, 1);
        return -2;
    }
    if (tile == 6) {
        if (ln2) return -2;
        if (nst == 2) OSG_W(64, 80, 2, false, 0, 5, 1);
        if (nst == 4) OSG_W(64, 80, 4, false, 0, 5, 1);
        if (nst == 6) OSG_W(64, 80, 6, false, 0, 5, 1);
        return -2;
    }
    if (tile == 7) {
        if (ln2) { if (nst == 2) OSG_W_LN2(64, 160, 2, 2); if (nst == 4) OSG_W_LN2(64, 160, 4, 2); return -2; }
        if (nst == 2) OSG_W(64, 160, 2, false, 0, 5, 2);
        if (nst == 4) OSG_W(64, 160, 4, false, 0, 5, 2);
        return -2;
    }
#undef OSG_W_LN2
#undef OSG_W
    return -2;
}

// does launch_v2_wide hold this (tile, ring, form)?  (the candidate generator of osg_gemm.hip and the table loader of osg_ctx.hip ask)
bool wide_tile_has(int tile, int nst, bool conv, bool ln1, bool ln2, bool geglu, bool rowstats) {
    if (ln1 || tile < 4 || tile > 7) return false;
    if ((geglu || rowstats) && tile != 4) return false;
    if (conv) return !ln2 && nst == 4 && tile != 7;
    if (ln2 && (tile == 5 || tile == 6)) return false;
    if (tile == 6) return nst == 2 || nst == 4 || nst == 6;
    return nst == 2 || nst == 4;
}

}  // namespace osg_mm
