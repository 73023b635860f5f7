// Internal definitions shared by the HIP translation units of libosgpu (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <string>
#include <map>
#include <vector>
#include "osgpu.h"

struct osg_ctx {
    int device = 0;
    hipStream_t compute = nullptr;
    hipStream_t copy = nullptr;
    hipStream_t copy2 = nullptr;        // second H2D queue of the streamed-weights pass (osg_upload_pinned_async alternates; OSG_COPY_STREAMS=1: off)
    hipEvent_t ev_copy2 = nullptr;
    int copy_streams = 2, copy_rr = 0;
    bool copy_dirty[2] = {false, false};
    hipEvent_t ev_copy = nullptr;       // copy stream -> compute stream dependency
    // GroupNorm statistics sinks (osg_set_stat_sinks): taken by the next osg_conv2d_nhwc_v; sink_fused = the launch that ran served them in its epilogue
    struct PendingSink { long long* table = nullptr; int groups = 0, cpg = 0, ch_off = 0; } pending_sink[2];
    int pending_hw = 0;
    bool tuning = false, sink_fused = false;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    // pinned double-buffered staging for host->device streaming (weights provider path)
    static constexpr int kStages = 2;
    void* stage[kStages] = {nullptr, nullptr};
    hipEvent_t stage_free[kStages] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    int stage_next = 0;
    // split-K / reduction workspace (grown on demand, never inside a capture)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    void* ws2 = nullptr;                // scratch for re-laid-out dynamic GEMM operands
    size_t ws2_bytes = 0;
    static constexpr long kTickets = 1 << 16;
    int* tickets = nullptr;             // split-K arrival counters (zeroed once; the last arriver of a tile resets its counter)
    bool xcd_ids8 = false;              // the device has exactly 8 XCDs whose XCC_ID are 0..7 (StatSink tables: one copy per XCD, L2-scope atomics)
    int* xcd_err = nullptr;             // host pointer
    int* xcd_err_dev = nullptr;         // the same word as the device sees it
    bool capturing = false;
    bool autotune = false;              // osg_set_autotune: contraction launches pick tile/split configurations by measurement (osg_tune.h)
    hipEvent_t ev_a0 = nullptr, ev_a1 = nullptr;
    void* evict = nullptr;              // OSG_TUNE_COLD: fill target that evicts L2 / MALL before a timed launch
    std::vector<hipEvent_t> marks;      // osg_timer_mark / osg_timer_between
    static constexpr int kMarkers = 256;
    hipEvent_t markers[kMarkers] = {};   // osg_marker_record / osg_copy_wait_marker (created on first use)
    std::string err;
    std::string name;
    int num_cu = 256;
};

// memo "this kernel's attributes are set" per DEVICE (advisor, round 5: a static bool left the large-LDS opt-in unset on a second device of the process)
inline bool osg_first_on_device(unsigned long long& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    if ((mask >> dev) & 1ull) return false;
    mask |= 1ull << dev;
    return true;
}

struct osg_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

#define OSG_FAIL(ctx, msg)                 \
    do {                                   \
        (ctx)->err = (msg);                \
        return 1;                          \
    } while (0)

#define OSG_HIP(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

#define OSG_LAUNCH_CHECK(ctx)                                                                     \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e__);                 \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

int osg_ensure_workspace(osg_ctx* ctx, size_t bytes);
int osg_ensure_workspace2(osg_ctx* ctx, size_t bytes);

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 1 / (1 + e^-x) on the hardware transcendentals (v_exp_f32 + v_rcp_f32, ~1 ulp each): the reference's XNNPACK f16 sigmoid is itself an
// approximation; saturates correctly (exp2 -> inf -> rcp -> 0).  One definition for every kernel, so fused and unfused
// paths produce the same bits.
__device__ __forceinline__ float osg_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
// 0.5 g (1 + erf(g / sqrt 2)) for the fused GEGLU paths: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7 + fp32 evaluation, i.e. far
// below the f16 rounding of the product) on v_rcp_f32 / v_exp_f32 -- a dozen instructions where libm's erff takes ~50; the GEGLU
// epilogue evaluates it for every output element of the largest GEMMs of the net.  The standalone Erf op keeps erff (the reference
// rounds std::erf's result, onnxstream.cpp:4001-4139).
__device__ __forceinline__ float osg_gelu_erf(float g) {
    const float xs = g * 0.70710678118654752440f, ax = fabsf(xs);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896341f);
    const float er = copysignf(fmaf(-poly, e, 1.0f), xs);
    return 0.5f * g * (1.0f + er);
}
__device__ __forceinline__ float osg_apply_act(float v, int act) {
    if (act == OSG_ACT_SILU) return v * osg_sigmoid(v);
    if (act == OSG_ACT_SIGMOID) return osg_sigmoid(v);
    return v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<f16>(f16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }  // v_cvt_f16_f32: RNE
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }

// Kernel arguments pulled into scalar registers NOW, in one batch (round 6).  hipcc loads a kernel argument where it is first used: a kernel's prologue then walks
// 3-7 DEPENDENT round trips to the kernel-argument segment (~0.1-0.4 us each inside a pass, where nothing is warm) before its first vector load -- and a launch of the
// pass lasts as long as one workgroup.  An INPUT of an empty asm makes the value live at the top, so the loads of all arguments are issued together; the values stay
// what they were to the optimiser (an in-out operand would hide a pointer's address space: flat loads).
template <class T>
__device__ __forceinline__ void osg_pin_one(const T& v) { asm volatile("" ::"s"(v)); }
template <class... T>
__device__ __forceinline__ void osg_pin_all(const T&... v) { (osg_pin_one(v), ...); }

// wave-wide sum on the VALU (round 6, GroupNorm's slab kernel): inside each 16-lane row by DPP (quad butterfly, then row rotations), across the four rows by the gfx950
// v_permlane16_swap / v_permlane32_swap pair (see osg_attention.hip: lane_swap16 / lane_swap32) -- ~10 VALU instructions where the ds_bpermute chain of wave_sum
// below is six LDS round trips of ~100 cycles.  Every lane ends up with the total; the additions are a fixed tree (another one than wave_sum's: different last bits).
template <int CTRL>
__device__ __forceinline__ float osg_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_valu(float v) {
    v += osg_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    v += osg_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    v += osg_dpp<0x124>(v);    // row_ror:4
    v += osg_dpp<0x128>(v);    // row_ror:8
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
