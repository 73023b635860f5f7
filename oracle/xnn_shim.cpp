// TEST INFRASTRUCTURE ONLY (oracle). Never linked into the product libraries.
//
// xnn_shim.cpp -- supplies the 30 XNNPACK entry points that the reference
// (/root/reference/src/onnxstream.cpp) calls but that the XNNPACK build bundled in
// torch/lib/libtorch_cpu.so does not export.  With this shim the
// UNMODIFIED reference source links against torch's XNNPACK and becomes the parity oracle
// (oracle/_ref/libonnxstream_ref.so).
//
// Design:
//  * "operator style" entry points (softmax, dynamic fully-connected, SDPA, nchw conv) hand back a
//    fake xnn_operator_t that starts with a 16-byte magic tag; this file interposes
//    xnn_run_operator / xnn_delete_operator and forwards every untagged operator to the genuine
//    XNNPACK implementation found with dlsym(RTLD_NEXT).
//  * dynamic fully-connected keeps the GEMM arithmetic inside genuine XNNPACK: at run time it
//    creates a *static* xnn fully_connected_nc_{f16,f32} operator with XNN_FLAG_TRANSPOSE_WEIGHTS
//    (kernel given as [K,N] row-major, as reference onnxstream.cpp:977 requests), runs and deletes it.
//    => f16 inputs, f32 accumulation, f16 output (xnn_f16_f32acc_gemm micro-kernels).
//  * softmax f16/f32 restates XNNPACK's three-pass algorithm (rmax, raddstoreexpminusmax, vmulc):
//    exp(x-max) is evaluated in f32, stored in the output precision, the sum is accumulated in f32 from
//    the un-rounded values, and the output is e * (1/sum) with the reciprocal rounded to the output
//    precision first.  Differences to the pinned XNNPACK build are of 1-fp16-ulp class (polynomial exp).
//  * transpose_nd is a plain N-d index walk, parallelised over the outermost output dimension.
//  * SDPA / nchw-conv are not on the SD hot path: they report xnn_status_unsupported_hardware; qu8 softmax (the W8A8 VAE's attention)
//    restates XNNPACK's lookup-table operator
//    so that the reference throws its normal "failed to create ..." exception.
#include <xnnpack.h>
#include <pthreadpool.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#include <immintrin.h>

namespace {

const unsigned char kMagic[16] = {0x4f, 0x53, 0x47, 0x53, 0x48, 0x49, 0x4d, 0x21,
                                  0x9a, 0x17, 0xc3, 0x5e, 0x00, 0xd1, 0x7b, 0xe2};

enum class Kind { softmax_f16, softmax_f32, softmax_qu8, dynfc_f16, dynfc_f32 };

struct FakeOp {
    unsigned char magic[16];
    Kind kind;
    // softmax
    size_t channels = 0, in_stride = 0, out_stride = 0, batch = 0;
    float qu8_in_scale = 0.f;           // qu8: lookup table of scaled exp((i - 255) * input_scale), built at reshape (depends on channels)
    uint32_t lut[256] = {};
    const void* in = nullptr;
    void* out = nullptr;
    // dynamic fc
    float out_min = 0, out_max = 0;
    uint32_t flags = 0;
    size_t M = 0, K = 0, N = 0, fc_in_stride = 0, fc_out_stride = 0;
    const void* kernel = nullptr;
    const void* bias = nullptr;
};

bool is_fake(xnn_operator_t op) { return op && std::memcmp((void*)op, kMagic, 16) == 0; }

FakeOp* make_fake(Kind k) {
    FakeOp* f = new FakeOp();
    std::memcpy(f->magic, kMagic, 16);
    f->kind = k;
    return f;
}

typedef enum xnn_status (*run_fn)(xnn_operator_t, pthreadpool_t);
typedef enum xnn_status (*del_fn)(xnn_operator_t);

run_fn real_run() {
    static run_fn f = (run_fn)dlsym(RTLD_NEXT, "xnn_run_operator");
    return f;
}
del_fn real_del() {
    static del_fn f = (del_fn)dlsym(RTLD_NEXT, "xnn_delete_operator");
    return f;
}

inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
inline uint16_t f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }

struct SoftmaxCtx { FakeOp* op; };

// qu8 softmax, restating XNNPACK's operator (src/operators/softmax-nc.c + u8-rmax + u8-lut32norm scalar microkernel, pinned commit
// google/XNNPACK@5671db05): t[i] = lrint(qscale * exp((i - 255) * input_scale)), qscale = min(UINT32_MAX / channels, 2^23 - 1);
// per row: m = max(x); sum = SUM t[x + 255 - m]; y = min(255, ((t[x + 255 - m] << 8) + (sum >> 1)) / sum).  Output scale is 1/256 with
// zero point 0 (the only combination XNNPACK accepts).  Integer arithmetic after the table: bit-exact by construction.
void softmax_row_qu8(void* vctx, size_t row) {
    FakeOp* f = ((SoftmaxCtx*)vctx)->op;
    const uint8_t* x = (const uint8_t*)f->in + row * f->in_stride;
    uint8_t* y = (uint8_t*)f->out + row * f->out_stride;
    uint8_t m = 0;
    for (size_t i = 0; i < f->channels; i++) m = x[i] > m ? x[i] : m;
    const uint32_t* t = f->lut + (255 - m);
    uint32_t sum = 0;
    for (size_t i = 0; i < f->channels; i++) sum += t[x[i]];
    const uint32_t rounding = sum >> 1;
    for (size_t i = 0; i < f->channels; i++) {
        const uint32_t q = (uint32_t)((((uint64_t)t[x[i]] << 8) + rounding) / sum);
        y[i] = q > 255 ? 255 : (uint8_t)q;
    }
}

void softmax_row_f16(void* vctx, size_t row) {
    FakeOp* op = ((SoftmaxCtx*)vctx)->op;
    const uint16_t* x = (const uint16_t*)op->in + row * op->in_stride;
    uint16_t* y = (uint16_t*)op->out + row * op->out_stride;
    const size_t n = op->channels;
    float mx = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < n; i++) mx = std::fmax(mx, h2f(x[i]));
    float sum = 0.f;
    for (size_t i = 0; i < n; i++) {
        float e = std::exp(h2f(x[i]) - mx);
        y[i] = f2h(e);
        sum += e;
    }
    const float rinv = h2f(f2h(1.0f / sum));
    for (size_t i = 0; i < n; i++) y[i] = f2h(h2f(y[i]) * rinv);
}

void softmax_row_f32(void* vctx, size_t row) {
    FakeOp* op = ((SoftmaxCtx*)vctx)->op;
    const float* x = (const float*)op->in + row * op->in_stride;
    float* y = (float*)op->out + row * op->out_stride;
    const size_t n = op->channels;
    float mx = -std::numeric_limits<float>::infinity();
    for (size_t i = 0; i < n; i++) mx = std::fmax(mx, x[i]);
    float sum = 0.f;
    for (size_t i = 0; i < n; i++) {
        float e = std::exp(x[i] - mx);
        y[i] = e;
        sum += e;
    }
    const float rinv = 1.0f / sum;
    for (size_t i = 0; i < n; i++) y[i] = y[i] * rinv;
}

enum xnn_status run_dynfc(FakeOp* f, pthreadpool_t tp) {
    xnn_operator_t op = nullptr;
    enum xnn_status st;
    const bool f16 = f->kind == Kind::dynfc_f16;
    if (f16)
        st = xnn_create_fully_connected_nc_f16(f->K, f->N, f->fc_in_stride, f->fc_out_stride, f->kernel, f->bias,
                                               f->out_min, f->out_max, f->flags, nullptr, nullptr, &op);
    else
        st = xnn_create_fully_connected_nc_f32(f->K, f->N, f->fc_in_stride, f->fc_out_stride, (const float*)f->kernel,
                                               (const float*)f->bias, f->out_min, f->out_max, f->flags, nullptr, nullptr, &op);
    if (st != xnn_status_success) return st;
    st = f16 ? xnn_reshape_fully_connected_nc_f16(op, f->M, tp) : xnn_reshape_fully_connected_nc_f32(op, f->M, tp);
    if (st == xnn_status_success)
        st = f16 ? xnn_setup_fully_connected_nc_f16(op, f->in, f->out)
                 : xnn_setup_fully_connected_nc_f32(op, (const float*)f->in, (float*)f->out);
    if (st == xnn_status_success) st = real_run()(op, tp);
    real_del()(op);
    return st;
}

// ---- transpose ------------------------------------------------------------------------------
struct TrCtx {
    const unsigned char* in;
    unsigned char* out;
    size_t nd;
    size_t out_shape[8];
    size_t in_stride_for_out[8];  // element stride in the input for each output dim
    size_t inner_out;             // elements per outermost-output index
    size_t esz;
};

template <typename T>
void tr_task(void* vctx, size_t o0) {
    TrCtx* c = (TrCtx*)vctx;
    const T* in = (const T*)c->in;
    T* out = (T*)c->out + o0 * c->inner_out;
    const size_t nd = c->nd;
    const size_t base = o0 * c->in_stride_for_out[0];
    const size_t last = c->out_shape[nd - 1];
    const size_t ls = c->in_stride_for_out[nd - 1];
    const size_t rows = c->inner_out / last;
    for (size_t r = 0; r < rows; r++) {
        size_t off = base, rem = r;
        for (size_t d = nd - 1; d-- > 1;) {
            off += (rem % c->out_shape[d]) * c->in_stride_for_out[d];
            rem /= c->out_shape[d];
        }
        const T* src = in + off;
        if (ls == 1)
            std::memcpy(out, src, last * sizeof(T));
        else
            for (size_t i = 0; i < last; i++) out[i] = src[i * ls];
        out += last;
    }
}

template <typename T>
enum xnn_status transpose_nd(const void* input, void* output, size_t nd, const size_t* shape, const size_t* perm,
                             pthreadpool_t tp) {
    if (nd == 0 || nd > 8) return xnn_status_invalid_parameter;
    TrCtx c;
    c.in = (const unsigned char*)input;
    c.out = (unsigned char*)output;
    c.esz = sizeof(T);
    size_t in_strides[8];
    size_t s = 1;
    for (size_t d = nd; d-- > 0;) { in_strides[d] = s; s *= shape[d]; }
    if (nd == 1) {
        std::memcpy(output, input, shape[0] * sizeof(T));
        return xnn_status_success;
    }
    c.nd = nd;
    size_t inner = 1;
    for (size_t d = 0; d < nd; d++) {
        c.out_shape[d] = shape[perm[d]];
        c.in_stride_for_out[d] = in_strides[perm[d]];
        if (d > 0) inner *= c.out_shape[d];
    }
    c.inner_out = inner;
    pthreadpool_parallelize_1d(tp, (pthreadpool_task_1d_t)tr_task<T>, &c, c.out_shape[0], 0);
    return xnn_status_success;
}

}  // namespace

extern "C" {

// ---- interposed generic entry points ----------------------------------------------------------
enum xnn_status xnn_run_operator(xnn_operator_t op, pthreadpool_t tp) {
    if (!is_fake(op)) return real_run()(op, tp);
    FakeOp* f = (FakeOp*)op;
    switch (f->kind) {
        case Kind::softmax_f16: {
            SoftmaxCtx c{f};
            pthreadpool_parallelize_1d(tp, softmax_row_f16, &c, f->batch, 0);
            return xnn_status_success;
        }
        case Kind::softmax_f32: {
            SoftmaxCtx c{f};
            pthreadpool_parallelize_1d(tp, softmax_row_f32, &c, f->batch, 0);
            return xnn_status_success;
        }
        case Kind::softmax_qu8: {
            SoftmaxCtx c{f};
            pthreadpool_parallelize_1d(tp, softmax_row_qu8, &c, f->batch, 0);
            return xnn_status_success;
        }
        case Kind::dynfc_f16:
        case Kind::dynfc_f32:
            return run_dynfc(f, tp);
    }
    return xnn_status_invalid_state;
}

enum xnn_status xnn_delete_operator(xnn_operator_t op) {
    if (!is_fake(op)) return real_del()(op);
    delete (FakeOp*)op;
    return xnn_status_success;
}

// ---- softmax -----------------------------------------------------------------------------------
enum xnn_status xnn_create_softmax_nc_f16(uint32_t, xnn_operator_t* out) {
    *out = (xnn_operator_t)make_fake(Kind::softmax_f16);
    return xnn_status_success;
}
enum xnn_status xnn_create_softmax_nc_f32(uint32_t, xnn_operator_t* out) {
    *out = (xnn_operator_t)make_fake(Kind::softmax_f32);
    return xnn_status_success;
}
static enum xnn_status reshape_softmax(xnn_operator_t op, size_t channels, size_t is, size_t os, size_t batch) {
    if (!is_fake(op)) return xnn_status_invalid_parameter;
    FakeOp* f = (FakeOp*)op;
    f->channels = channels; f->in_stride = is; f->out_stride = os; f->batch = batch;
    return xnn_status_success;
}
enum xnn_status xnn_reshape_softmax_nc_f16(xnn_operator_t op, size_t c, size_t is, size_t os, size_t b, pthreadpool_t) {
    return reshape_softmax(op, c, is, os, b);
}
enum xnn_status xnn_reshape_softmax_nc_f32(xnn_operator_t op, size_t c, size_t is, size_t os, size_t b, pthreadpool_t) {
    return reshape_softmax(op, c, is, os, b);
}
enum xnn_status xnn_setup_softmax_nc_f16(xnn_operator_t op, const void* in, void* out) {
    FakeOp* f = (FakeOp*)op; f->in = in; f->out = out; return xnn_status_success;
}
enum xnn_status xnn_setup_softmax_nc_f32(xnn_operator_t op, const float* in, float* out) {
    FakeOp* f = (FakeOp*)op; f->in = in; f->out = out; return xnn_status_success;
}
enum xnn_status xnn_create_softmax_nc_qu8(float input_scale, uint8_t output_zero_point, float output_scale, uint32_t, xnn_operator_t* out) {
    *out = nullptr;
    if (!(input_scale > 0.f) || output_zero_point != 0 || output_scale != 0x1.0p-8f) return xnn_status_unsupported_parameter;   // XNNPACK's own gate
    FakeOp* f = make_fake(Kind::softmax_qu8);
    f->qu8_in_scale = input_scale;
    *out = (xnn_operator_t)f;
    return xnn_status_success;
}
enum xnn_status xnn_reshape_softmax_nc_qu8(xnn_operator_t op, size_t c, size_t is, size_t os, size_t b, pthreadpool_t) {
    if (!is_fake(op) || c == 0) return xnn_status_invalid_parameter;
    FakeOp* f = (FakeOp*)op;
    const double qscale = std::fmin(((double)UINT32_MAX) / (double)c, 8388607.0);
    for (int i = 0; i < 256; i++) f->lut[i] = (uint32_t)std::lrint(qscale * std::exp((double)(i - 255) * (double)f->qu8_in_scale));
    return reshape_softmax(op, c, is, os, b);
}
enum xnn_status xnn_setup_softmax_nc_qu8(xnn_operator_t op, const uint8_t* in, uint8_t* out) {
    FakeOp* f = (FakeOp*)op; f->in = in; f->out = out; return xnn_status_success;
}

// ---- dynamic fully connected -------------------------------------------------------------------
static enum xnn_status create_dynfc(Kind k, float mn, float mx, uint32_t flags, xnn_operator_t* out) {
    FakeOp* f = make_fake(k);
    f->out_min = mn; f->out_max = mx; f->flags = flags;
    *out = (xnn_operator_t)f;
    return xnn_status_success;
}
enum xnn_status xnn_create_dynamic_fully_connected_nc_f16(float mn, float mx, uint32_t flags, xnn_operator_t* out) {
    return create_dynfc(Kind::dynfc_f16, mn, mx, flags, out);
}
enum xnn_status xnn_create_dynamic_fully_connected_nc_f32(float mn, float mx, uint32_t flags, xnn_operator_t* out) {
    return create_dynfc(Kind::dynfc_f32, mn, mx, flags, out);
}
static enum xnn_status reshape_dynfc(xnn_operator_t op, size_t batch, size_t ic, size_t oc, size_t is, size_t os,
                                     size_t* ws, size_t* wa) {
    if (!is_fake(op)) return xnn_status_invalid_parameter;
    FakeOp* f = (FakeOp*)op;
    f->M = batch; f->K = ic; f->N = oc; f->fc_in_stride = is; f->fc_out_stride = os;
    if (ws) *ws = 0;
    if (wa) *wa = 1;
    return xnn_status_success;
}
enum xnn_status xnn_reshape_dynamic_fully_connected_nc_f16(xnn_operator_t op, size_t b, size_t ic, size_t oc, size_t is,
                                                           size_t os, size_t* ws, size_t* wa, pthreadpool_t) {
    return reshape_dynfc(op, b, ic, oc, is, os, ws, wa);
}
enum xnn_status xnn_reshape_dynamic_fully_connected_nc_f32(xnn_operator_t op, size_t b, size_t ic, size_t oc, size_t is,
                                                           size_t os, size_t* ws, size_t* wa, pthreadpool_t) {
    return reshape_dynfc(op, b, ic, oc, is, os, ws, wa);
}
enum xnn_status xnn_setup_dynamic_fully_connected_nc_f16(xnn_operator_t op, void*, const void* in, const void* kernel,
                                                         const void* bias, void* out) {
    FakeOp* f = (FakeOp*)op; f->in = in; f->kernel = kernel; f->bias = bias; f->out = out; return xnn_status_success;
}
enum xnn_status xnn_setup_dynamic_fully_connected_nc_f32(xnn_operator_t op, void*, const float* in, const float* kernel,
                                                         const float* bias, float* out) {
    FakeOp* f = (FakeOp*)op; f->in = in; f->kernel = kernel; f->bias = bias; f->out = out; return xnn_status_success;
}

// ---- transpose -----------------------------------------------------------------------------------
enum xnn_status xnn_run_transpose_nd_x8(const void* in, void* out, size_t nd, const size_t* shape, const size_t* perm,
                                        uint32_t, pthreadpool_t tp) {
    return transpose_nd<uint8_t>(in, out, nd, shape, perm, tp);
}
enum xnn_status xnn_run_transpose_nd_x16(const void* in, void* out, size_t nd, const size_t* shape, const size_t* perm,
                                         uint32_t, pthreadpool_t tp) {
    return transpose_nd<uint16_t>(in, out, nd, shape, perm, tp);
}
enum xnn_status xnn_run_transpose_nd_x32(const void* in, void* out, size_t nd, const size_t* shape, const size_t* perm,
                                         uint32_t, pthreadpool_t tp) {
    return transpose_nd<uint32_t>(in, out, nd, shape, perm, tp);
}

// ---- not on the SD hot path: unsupported ---------------------------------------------------------
enum xnn_status xnn_create_scaled_dot_product_attention_nhtc_f16(enum xnn_attention_logits_cap_type, const void*, uint32_t,
                                                                 xnn_operator_t* out) {
    *out = nullptr; return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_create_scaled_dot_product_attention_nhtc_f32(enum xnn_attention_logits_cap_type, const void*, uint32_t,
                                                                 xnn_operator_t* out) {
    *out = nullptr; return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_reshape_scaled_dot_product_attention_nhtc_f16(xnn_operator_t, size_t, size_t, size_t, size_t, size_t,
                                                                  size_t, size_t, size_t*, size_t*, pthreadpool_t) {
    return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_reshape_scaled_dot_product_attention_nhtc_f32(xnn_operator_t, size_t, size_t, size_t, size_t, size_t,
                                                                  size_t, size_t, size_t*, size_t*, pthreadpool_t) {
    return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_setup_scaled_dot_product_attention_nhtc_f16(xnn_operator_t, void*, const void*, const void*, const void*,
                                                                const void*, const void*, void*) {
    return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_setup_scaled_dot_product_attention_nhtc_f32(xnn_operator_t, void*, const float*, const float*,
                                                                const float*, const float*, const float*, float*) {
    return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_create_convolution2d_nchw_f16(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                                                  uint32_t, uint32_t, uint32_t, uint32_t, size_t, size_t, size_t, size_t,
                                                  const void*, const void*, float, float, uint32_t, xnn_code_cache_t,
                                                  xnn_weights_cache_t, xnn_operator_t* out) {
    *out = nullptr; return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_create_convolution2d_nchw_f32(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                                                  uint32_t, uint32_t, uint32_t, uint32_t, size_t, size_t, size_t, size_t,
                                                  const float*, const float*, float, float, uint32_t, xnn_code_cache_t,
                                                  xnn_weights_cache_t, xnn_operator_t* out) {
    *out = nullptr; return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_reshape_convolution2d_nchw_f16(xnn_operator_t, size_t, size_t, size_t, size_t*, size_t*, pthreadpool_t) {
    return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_reshape_convolution2d_nchw_f32(xnn_operator_t, size_t, size_t, size_t, size_t*, size_t*, pthreadpool_t) {
    return xnn_status_unsupported_hardware;
}
enum xnn_status xnn_setup_convolution2d_nchw_f16(xnn_operator_t, const void*, void*) { return xnn_status_unsupported_hardware; }
enum xnn_status xnn_setup_convolution2d_nchw_f32(xnn_operator_t, const float*, float*) { return xnn_status_unsupported_hardware; }

}  // extern "C"
