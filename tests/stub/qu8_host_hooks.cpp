// TEST INFRASTRUCTURE ONLY -- extern "C" wrappers over onnxstream_amd/csrc/host/qu8.h so that tests/test_qu8_host.py can compare the host-side
// uint8 parameter builders with oracle/np_qu8.py (compiled by the test itself; never part of a product library).
#include "qu8.h"
using namespace onnxstream::qu8;
extern "C" {
void h_range_to_scale(float lo, float hi, float* scale, int* zp) { auto q = range_to_scale(lo, hi); *scale = q.scale; *zp = q.zero_point; }
int h_quantize_dynamic(const float* x, size_t n, size_t threads, uint8_t* out, float* scale, int* zp) {
    QParams q{};
    if (!quantize_dynamic(x, n, threads, out, &q)) return 1;
    *scale = q.scale; *zp = q.zero_point;
    return 0;
}
int h_percentiles(const float* x, size_t n, size_t threads, size_t workers, float* lo, float* hi) {
    auto r = workers ? percentiles_fast(x, n, 0.001f, 0.001f, threads, workers) : percentiles(x, n, 0.001f, 0.001f, threads);
    if (!r) return 1;
    *lo = r->first; *hi = r->second;
    return 0;
}
void h_sigmoid_lut(float si, int zi, float so, int zo, uint8_t* lut) { sigmoid_lut({si, (uint8_t)zi}, {so, (uint8_t)zo}, lut); }
void h_add(const uint8_t* a, float sa, int za, const uint8_t* b, float sb, int zb, float so, int zo, size_t n, uint8_t* y) {
    const AddParams p = add_params({sa, (uint8_t)za}, {sb, (uint8_t)zb}, {so, (uint8_t)zo});
    for (size_t i = 0; i < n; i++) y[i] = add(a[i], b[i], p);
}
void h_mul(const uint8_t* a, float sa, int za, const uint8_t* b, float sb, int zb, float so, int zo, size_t n, uint8_t* y) {
    const float sc = requant_scale(sa, sb, so);
    for (size_t i = 0; i < n; i++) y[i] = requant_fp32(((int32_t)a[i] - za) * ((int32_t)b[i] - zb), sc, (uint8_t)zo);
}
void h_requant(const int32_t* acc, float sa, float sb, float so, int zo, size_t n, uint8_t* y) {
    const float sc = requant_scale(sa, sb, so);
    for (size_t i = 0; i < n; i++) y[i] = requant_fp32(acc[i], sc, (uint8_t)zo);
}
void h_conv_bias(const float* b, float sx, float sw, size_t n, int32_t* out) { for (size_t i = 0; i < n; i++) out[i] = conv_bias_i32(b[i], sx, sw); }
void h_softmax(const uint8_t* x, float s_in, size_t rows, size_t channels, uint8_t* y) {
    uint32_t lut[256];
    softmax_lut(s_in, channels, lut);
    for (size_t r = 0; r < rows; r++) softmax_row(x + r * channels, channels, lut, y + r * channels);
}
void h_instance_norm(const uint8_t* x, size_t C, size_t L, float si, int zi, const float* scale, const float* bias, float eps, float so, int zo, uint8_t* y) {
    for (size_t c = 0; c < C; c++) {
        uint32_t hist[256] = {};
        for (size_t i = 0; i < L; i++) hist[x[c * L + i]]++;
        uint8_t lut[256];
        instance_norm_lut(hist, L, {si, (uint8_t)zi}, scale[c], bias[c], eps, {so, (uint8_t)zo}, lut);
        for (size_t i = 0; i < L; i++) y[c * L + i] = lut[x[c * L + i]];
    }
}
}
