// libosgpu: elementwise / broadcast / conversion kernels (HBM-bound: 16-byte vector accesses, grid-stride).
// Replaces XnnPack::{add,subtract,multiply,divide,sigmoid,convert,convert_qu8} (reference onnxstream.cpp:846-1957)
// and the inline Erf/Sqrt/Sin/Cos/Pow/Neg host loops of Model::run (:4001-4139, :5478-5604, :7475).
// Arithmetic contract: operands widened to f32, op in f32, one RNE rounding to the storage type.
#include "osg_common.h"

namespace {

constexpr int kMaxRank = 6;

__device__ __forceinline__ float apply_unary(float x, int kind, float param) {
    switch (kind) {
        case OSG_UN_SIGMOID: return osg_sigmoid(x);
        case OSG_UN_ERF: return erff(x);
        case OSG_UN_SQRT: return sqrtf(x);
        case OSG_UN_SIN: return sinf(x);
        case OSG_UN_COS: return cosf(x);
        case OSG_UN_NEG: return -x;
        case OSG_UN_POW: return param == 2.0f ? x * x : powf(x, param);
        case OSG_UN_SILU: return x * osg_sigmoid(x);
        case OSG_UN_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
    }
    return x;
}

__device__ __forceinline__ float apply_binary(float a, float b, int kind) {
    switch (kind) {
        case OSG_BIN_ADD: return a + b;
        case OSG_BIN_SUB: return a - b;
        case OSG_BIN_MUL: return a * b;
        default: return a / b;
    }
}

template <typename T, int VEC>
struct Vec;
template <> struct Vec<f16, 8> { typedef f16x8 type; };
template <> struct Vec<float, 4> { typedef f32x4 type; };

template <typename T> struct VW;
template <> struct VW<f16> { static constexpr int n = 8; };
template <> struct VW<float> { static constexpr int n = 4; };

template <typename T>
__global__ __launch_bounds__(256) void unary_kernel(const T* __restrict__ x, T* __restrict__ y, long n, int kind, float param) {
    constexpr int V = VW<T>::n;
    typedef typename Vec<T, V>::type vec_t;
    long nv = n / V;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        vec_t v = reinterpret_cast<const vec_t*>(x)[i];
        vec_t o;
#pragma unroll
        for (int e = 0; e < V; e++) o[e] = from_f32<T>(apply_unary(to_f32<T>(v[e]), kind, param));
        reinterpret_cast<vec_t*>(y)[i] = o;
    }
    long tail = nv * V + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tail < n) y[tail] = from_f32<T>(apply_unary(to_f32<T>(x[tail]), kind, param));
}

// same-shape, or b broadcast as a scalar / along the last dim (period `bper`, e.g. bias[C] over [rows,C] or the
// per-channel operand of an NHWC tensor)
template <typename T, int MODE>  // MODE 0: same shape, 1: b scalar, 2: b periodic (bper % V == 0), 3: a scalar, 4: a periodic
__global__ __launch_bounds__(256) void binary_fast_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n,
                                                          long per, int kind) {
    constexpr int V = VW<T>::n;
    typedef typename Vec<T, V>::type vec_t;
    long nv = n / V;
    long stride = (long)gridDim.x * blockDim.x;
    float sc = 0.f;
    if (MODE == 1) sc = to_f32<T>(b[0]);
    if (MODE == 3) sc = to_f32<T>(a[0]);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        vec_t va, vb, o;
        if (MODE == 3) {
#pragma unroll
            for (int e = 0; e < V; e++) va[e] = from_f32<T>(sc);
        } else if (MODE == 4) {
            va = *reinterpret_cast<const vec_t*>(a + (i * V) % per);
        } else {
            va = reinterpret_cast<const vec_t*>(a)[i];
        }
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < V; e++) vb[e] = from_f32<T>(sc);
        } else if (MODE == 2) {
            vb = *reinterpret_cast<const vec_t*>(b + (i * V) % per);
        } else {
            vb = reinterpret_cast<const vec_t*>(b)[i];
        }
#pragma unroll
        for (int e = 0; e < V; e++) o[e] = from_f32<T>(apply_binary(to_f32<T>(va[e]), to_f32<T>(vb[e]), kind));
        reinterpret_cast<vec_t*>(y)[i] = o;
    }
    long t = nv * V + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        float fa = MODE == 3 ? sc : (MODE == 4 ? to_f32<T>(a[t % per]) : to_f32<T>(a[t]));
        float fb = MODE == 1 ? sc : (MODE == 2 ? to_f32<T>(b[t % per]) : to_f32<T>(b[t]));
        y[t] = from_f32<T>(apply_binary(fa, fb, kind));
    }
}

struct BcastParams {
    long oshape[kMaxRank];
    long astride[kMaxRank];
    long bstride[kMaxRank];
    int rank;
};

template <typename T>
__global__ __launch_bounds__(256) void binary_bcast_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n,
                                                           BcastParams p, int kind) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long rem = i, ao = 0, bo = 0;
#pragma unroll
        for (int d = kMaxRank - 1; d >= 0; d--) {
            if (d < p.rank) {
                long q = rem / p.oshape[d];
                long idx = rem - q * p.oshape[d];
                rem = q;
                ao += idx * p.astride[d];
                bo += idx * p.bstride[d];
            }
        }
        y[i] = from_f32<T>(apply_binary(to_f32<T>(a[ao]), to_f32<T>(b[bo]), kind));
    }
}

// GEGLU: y[r, c] = x[r, c] * gelu_erf(x[r, C + c])
template <typename T>
__global__ __launch_bounds__(256) void geglu_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, long C) {
    constexpr int V = VW<T>::n;
    typedef typename Vec<T, V>::type vec_t;
    long cv = C / V;
    long nv = rows * cv;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        long r = i / cv, c = (i - r * cv) * V;
        vec_t val = *reinterpret_cast<const vec_t*>(x + r * 2 * C + c);
        vec_t gate = *reinterpret_cast<const vec_t*>(x + r * 2 * C + C + c);
        vec_t o;
#pragma unroll
        for (int e = 0; e < V; e++)
            o[e] = from_f32<T>(to_f32<T>(val[e]) * osg_gelu_erf(to_f32<T>(gate[e])));   // same expression as the GEMM's GEGLU epilogue
        *reinterpret_cast<vec_t*>(y + r * C + c) = o;
    }
}

// conversions.  Quantised formulas are the bit-exact contracts probed against the reference's XNNPACK:
//   u8 -> f : (float)((int)q - zp) * scale         f -> u8 : clamp(rne(x * (1.0f/scale)) + zp, 0, 255)
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ x, D* __restrict__ y, long n, float scale, int zp) {
    long stride = (long)gridDim.x * blockDim.x;
    const float inv = 1.0f / scale;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if constexpr (sizeof(S) == 1 && sizeof(D) != 1) {
            float f = (float)((int)x[i] - zp) * scale;
            y[i] = from_f32<D>(f);
        } else if constexpr (sizeof(D) == 1 && sizeof(S) != 1) {
            float f = to_f32<S>(x[i]);
            float r = rintf(f * inv) + (float)zp;
            r = fminf(fmaxf(r, 0.f), 255.f);
            y[i] = (uint8_t)r;
        } else if constexpr (sizeof(S) != 1 && sizeof(D) != 1) {
            y[i] = from_f32<D>(to_f32<S>(x[i]));
        } else {
            y[i] = (D)x[i];
        }
    }
}

inline unsigned grid_for(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (unsigned)blocks;
}

template <typename T>
int run_binary(osg_ctx* ctx, int kind, const T* a, const long* ash, const T* b, const long* bsh, T* y, int rank) {
    constexpr int V = VW<T>::n;
    long oshape[kMaxRank], an = 1, bn = 1, n = 1;
    for (int d = 0; d < rank; d++) {
        if (ash[d] != bsh[d] && ash[d] != 1 && bsh[d] != 1) OSG_FAIL(ctx, "osg_binary: shapes are not broadcastable");
        oshape[d] = ash[d] > bsh[d] ? ash[d] : bsh[d];
        an *= ash[d];
        bn *= bsh[d];
        n *= oshape[d];
    }
    if (n == 0) return 0;
    auto aligned = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const bool al = aligned(a) && aligned(b) && aligned(y);
    // trailing-suffix detection: operand equals the output on its last dims and is 1 elsewhere
    auto suffix_period = [&](const long* sh, long cnt) -> long {
        long per = 1;
        int d = rank - 1;
        for (; d >= 0 && sh[d] == oshape[d]; d--) per *= sh[d];
        for (; d >= 0; d--)
            if (sh[d] != 1) return 0;
        return per == cnt ? per : 0;
    };
#define OSG_FAST(MODE, PER)                                                                                              \
    do {                                                                                                                 \
        hipLaunchKernelGGL((binary_fast_kernel<T, MODE>), dim3(grid_for(n / V + 1)), dim3(256), 0, ctx->compute, a, b, y, n, \
                           (long)(PER), kind);                                                                           \
        OSG_LAUNCH_CHECK(ctx);                                                                                           \
        return 0;                                                                                                        \
    } while (0)
    if (al) {
        if (an == n && bn == n) OSG_FAST(0, 1);
        if (an == n && bn == 1) OSG_FAST(1, 1);
        if (bn == n && an == 1) OSG_FAST(3, 1);
        long pb = an == n ? suffix_period(bsh, bn) : 0;
        if (pb && pb % V == 0) OSG_FAST(2, pb);
        long pa = bn == n ? suffix_period(ash, an) : 0;
        if (pa && pa % V == 0) OSG_FAST(4, pa);
    }
#undef OSG_FAST
    BcastParams p{};
    p.rank = rank;
    long as = 1, bs = 1;
    for (int d = rank - 1; d >= 0; d--) {
        p.oshape[d] = oshape[d];
        p.astride[d] = ash[d] == 1 ? 0 : as;
        p.bstride[d] = bsh[d] == 1 ? 0 : bs;
        as *= ash[d];
        bs *= bsh[d];
    }
    hipLaunchKernelGGL(binary_bcast_kernel<T>, dim3(grid_for(n)), dim3(256), 0, ctx->compute, a, b, y, n, p, kind);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

__global__ __launch_bounds__(256) void rope_kernel(const f16* __restrict__ x, const f16* __restrict__ cs, const f16* __restrict__ sn, f16* __restrict__ y, long n,
                                                   long T, int d) {
#pragma clang fp contract(off)   // (the two products are rounded before the sum, as the separate Mul and Add ops round them: no fma across them)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i % d), half = d >> 1;
    const long t = (i / d) % T;
    const float xv = (float)x[i];
    const float rot = j < half ? -(float)x[i + half] : (float)x[i - half];
    const f16 a = (f16)(xv * (float)cs[t * d + j]);
    const f16 b = (f16)(rot * (float)sn[t * d + j]);
    y[i] = (f16)((float)a + (float)b);
}

}  // namespace

extern "C" {

// Rotary embedding as the LLM graphs spell it: y = x * cos + concat(-x[d/2:], x[:d/2]) * sin on [BH][T][d] with cos / sin [T][d] -- the seven ops
// Slice, Slice, Neg, Concat, Mul, Mul, Add in one launch with THEIR roundings (both products rounded to f16 before the sum): bit-identical
int osg_rope(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* cos_t, const void* sin_t, void* y, long bh, long T, int d) {
    if (bh <= 0 || T <= 0 || d <= 0) return 0;
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_rope: only f16 arithmetic is implemented on the device");
    if (d % 2) OSG_FAIL(ctx, "osg_rope: the head dim must be even");
    const long n = bh * T * d;
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->compute, (const f16*)x, (const f16*)cos_t, (const f16*)sin_t, (f16*)y, n, T, d);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_unary(osg_ctx* ctx, osg_dtype dtype, osg_unary_kind kind, const void* x, void* y, long n, float param) {
    if (n <= 0) return 0;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) OSG_FAIL(ctx, "osg_unary: buffers must be 16-byte aligned");
    if (dtype == OSG_F16)
        hipLaunchKernelGGL(unary_kernel<f16>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, ctx->compute, (const f16*)x, (f16*)y, n,
                           (int)kind, param);
    else if (dtype == OSG_F32)
        hipLaunchKernelGGL(unary_kernel<float>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, ctx->compute, (const float*)x, (float*)y,
                           n, (int)kind, param);
    else
        OSG_FAIL(ctx, "osg_unary: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_binary(osg_ctx* ctx, osg_dtype dtype, osg_binary_kind kind, const void* a, const long* a_shape, const void* b,
               const long* b_shape, void* y, int rank) {
    if (rank < 1 || rank > kMaxRank) OSG_FAIL(ctx, "osg_binary: rank must be in [1,6]");
    if (dtype == OSG_F16) return run_binary<f16>(ctx, kind, (const f16*)a, a_shape, (const f16*)b, b_shape, (f16*)y, rank);
    if (dtype == OSG_F32) return run_binary<float>(ctx, kind, (const float*)a, a_shape, (const float*)b, b_shape, (float*)y, rank);
    OSG_FAIL(ctx, "osg_binary: unsupported dtype");
}

int osg_geglu(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, long rows, long C) {
    if (rows <= 0 || C <= 0) return 0;
    if (dtype == OSG_F16) {
        if (C % 8) OSG_FAIL(ctx, "osg_geglu: C must be a multiple of 8");
        hipLaunchKernelGGL(geglu_kernel<f16>, dim3(grid_for(rows * C / 8)), dim3(256), 0, ctx->compute, (const f16*)x, (f16*)y, rows, C);
    } else if (dtype == OSG_F32) {
        if (C % 4) OSG_FAIL(ctx, "osg_geglu: C must be a multiple of 4");
        hipLaunchKernelGGL(geglu_kernel<float>, dim3(grid_for(rows * C / 4)), dim3(256), 0, ctx->compute, (const float*)x, (float*)y,
                           rows, C);
    } else
        OSG_FAIL(ctx, "osg_geglu: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_convert(osg_ctx* ctx, osg_dtype sd, osg_dtype dd, const void* x, void* y, long n, float scale, int zp) {
    if (n <= 0) return 0;
    dim3 g(grid_for(n)), b(256);
#define OSG_CVT(S, D) hipLaunchKernelGGL((convert_kernel<S, D>), g, b, 0, ctx->compute, (const S*)x, (D*)y, n, scale, zp)
    if (sd == OSG_F16 && dd == OSG_F32) OSG_CVT(f16, float);
    else if (sd == OSG_F32 && dd == OSG_F16) OSG_CVT(float, f16);
    else if (sd == OSG_F16 && dd == OSG_F16) OSG_CVT(f16, f16);
    else if (sd == OSG_F32 && dd == OSG_F32) OSG_CVT(float, float);
    else if (sd == OSG_U8 && dd == OSG_F32) OSG_CVT(uint8_t, float);
    else if (sd == OSG_U8 && dd == OSG_F16) OSG_CVT(uint8_t, f16);
    else if (sd == OSG_F32 && dd == OSG_U8) OSG_CVT(float, uint8_t);
    else if (sd == OSG_F16 && dd == OSG_U8) OSG_CVT(f16, uint8_t);
    else OSG_FAIL(ctx, "osg_convert: unsupported conversion");
#undef OSG_CVT
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
