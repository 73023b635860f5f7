// Measured configuration choice for the contraction launches (osg_gemm.hip, osg_conv3x3.hip).
//
// The tile / pipeline-depth / split-K choice of a GEMM or convolution is made by a small cost model (choose_v2, osg_conv3x3_run).  With
// osg_set_autotune(ctx, 1) the first EAGER launch of a shape instead times every configuration the model considers legal on the caller's
// own operands (HIP events on the compute stream, 1 warm + 2 timed launches each) and remembers the fastest; later launches -- including
// the ones captured into a hipGraph -- reuse it.  The table is process-wide and keyed by the device, so every context (every Model) of a
// process makes the same choice for the same shape: results stay bit-reproducible inside a process.  Contexts without autotune never
// consult the table.  OSG_TUNE_CACHE=<file> persists the table across processes.  Nothing is tuned during graph capture (no synchronisation is allowed there): the model's choice is used.
#pragma once
#include <cstdlib>
#include <tuple>
#include "osg_common.h"

namespace osg_tune {

struct Key {
    int kind;       // 0: GEMM, 1: 3x3/s1/p1 convolution (halo-reuse kernel and implicit GEMM compete), 2: other implicit-GEMM convolution
    int device;
    int M, N, K, batch;
    int H, W, Cin, KW, sh, sw;
    int flags;      // epilogue shape: act | residual << 4 | rowbias << 5 | bias_f32 << 6
    bool operator<(const Key& o) const {
        return std::tie(kind, device, M, N, K, batch, H, W, Cin, KW, sh, sw, flags) <
               std::tie(o.kind, o.device, o.M, o.N, o.K, o.batch, o.H, o.W, o.Cin, o.KW, o.sh, o.sw, o.flags);
    }
};

struct Choice {
    int family;     // 0: gemm2 (cfg, nst, splits); 1: conv3x3 (bn, splits)
    int cfg, nst, splits, bn;
    float us;       // measured time of the winner
};

bool lookup(const Key& k, Choice* out);
// OSG_TUNE_FROZEN=1: a shape the table does not hold is NOT timed -- it takes the cost model's first candidate (deterministic, nothing stored).  What the ranks of
// a multi-GPU job run with: every rank seeds the same shipped table and plans on its own, identically, without waiting for rank 0 to measure anything.
inline bool frozen() { static const bool v = getenv("OSG_TUNE_FROZEN") && atoi(getenv("OSG_TUNE_FROZEN")) != 0; return v; }
void store(const Key& k, const Choice& c);
void remember(const Key& k, const Choice& c);   // in memory only (the untimed choice of a shape a frozen table does not hold)
int misses();                                   // lookups that found nothing, since the process started

// microseconds per launch of f() (which enqueues the whole operation, reduce kernel included, and returns 0 on success); < 0 on failure
// cold timing (default; OSG_TUNE_COLD=0 switches it off): every timed launch starts with L2 / MALL evicted (a 384 MiB fill on the same stream before the first event) -- inside
// a pass the weights of a layer always come from HBM (1.7 GB stream through a 256 MB MALL), which back-to-back launches on one operand hide.
template <class F>
float time_us_impl(osg_ctx* ctx, F&& f);
// (the repetitions of a timed candidate are not launches of the pass: side effects that accumulate -- GroupNorm statistics sinks -- are off while ctx->tuning)
template <class F>
float time_us(osg_ctx* ctx, F&& f) {
    const bool was = ctx->tuning;
    ctx->tuning = true;
    const float us = time_us_impl(ctx, f);
    ctx->tuning = was;
    return us;
}
template <class F>
float time_us_impl(osg_ctx* ctx, F&& f) {
    // round 3: COLD timing is the default -- inside a pass every operand of a launch is cold (the previous launch wrote the activation, the weights were
    // last touched a pass ago), and candidates ranked on L2-hot operands favour shallow rings that then expose the full memory latency at every k-step:
    // the same bench on one box 6.44 (hot ranking) vs 6.17 ms per step (cold ranking), profiles/r03_tune_hot_vs_cold.txt.  OSG_TUNE_COLD=0: hot.
    static const bool cold = !getenv("OSG_TUNE_COLD") || atoi(getenv("OSG_TUNE_COLD")) != 0;
    if (f()) return -1.f;
    if (cold) {
        constexpr size_t kEvict = (size_t)384 << 20;
        if (!ctx->evict && hipMalloc(&ctx->evict, kEvict) != hipSuccess) return -1.f;
        float best = -1.f;
        for (int i = 0; i < 3; i++) {
            if (hipMemsetAsync(ctx->evict, i, kEvict, ctx->compute) != hipSuccess) return -1.f;
            if (hipEventRecord(ctx->ev_a0, ctx->compute) != hipSuccess) return -1.f;
            if (f()) return -1.f;
            if (hipEventRecord(ctx->ev_a1, ctx->compute) != hipSuccess) return -1.f;
            if (hipEventSynchronize(ctx->ev_a1) != hipSuccess) return -1.f;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->ev_a0, ctx->ev_a1) != hipSuccess) return -1.f;
            if (best < 0.f || ms * 1000.f < best) best = ms * 1000.f;
        }
        return best;
    }
    if (hipEventRecord(ctx->ev_a0, ctx->compute) != hipSuccess) return -1.f;
    if (f() || f()) return -1.f;
    if (hipEventRecord(ctx->ev_a1, ctx->compute) != hipSuccess) return -1.f;
    if (hipEventSynchronize(ctx->ev_a1) != hipSuccess) return -1.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev_a0, ctx->ev_a1) != hipSuccess) return -1.f;
    return ms * 500.f;
}

}  // namespace osg_tune
