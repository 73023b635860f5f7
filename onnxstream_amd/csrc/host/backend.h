// backend.h -- the host side of the drop-in boundary: class HipBackend takes the seat of the reference's
// `class XnnPack` (reference src/onnxstream.cpp:657-2150) inside Model, but instead of running XNNPACK operators on a
// pthreadpool it forwards to the C ABI of libosgpu.so (include/osgpu.h).  The library is bound at run time with dlopen
// so that libonnxstream_amd.so itself loads on a box without a GPU (symbol / parser tests); constructing a HipBackend
// without libosgpu.so or without a visible device throws -- there is no CPU fallback.
#pragma once

#include <dlfcn.h>

#include <stdexcept>
#include <string>

#include "osgpu.h"

namespace onnxstream {

struct OsgApi {
#define OSG_FN(name) decltype(&::name) name = nullptr;
    OSG_FN(osg_device_count) OSG_FN(osg_init) OSG_FN(osg_destroy) OSG_FN(osg_last_error) OSG_FN(osg_device_name) OSG_FN(osg_stream) OSG_FN(osg_set_autotune)
    OSG_FN(osg_malloc) OSG_FN(osg_free) OSG_FN(osg_upload) OSG_FN(osg_upload_sync) OSG_FN(osg_host_register) OSG_FN(osg_host_unregister) OSG_FN(osg_upload_pinned) OSG_FN(osg_download) OSG_FN(osg_copy)
    OSG_FN(osg_memset) OSG_FN(osg_sync) OSG_FN(osg_graph_begin) OSG_FN(osg_graph_end) OSG_FN(osg_graph_launch)
    OSG_FN(osg_graph_destroy) OSG_FN(osg_side_begin) OSG_FN(osg_side_end) OSG_FN(osg_side_join) OSG_FN(osg_timer_start) OSG_FN(osg_timer_stop) OSG_FN(osg_conv2d_nhwc) OSG_FN(osg_conv2d_nhwc_rb) OSG_FN(osg_gemm) OSG_FN(osg_gemm_ln) OSG_FN(osg_gemm_rowstats) OSG_FN(osg_gemm_w8) OSG_FN(osg_conv2d_nhwc_w8)
    OSG_FN(osg_transpose_kn_to_nk) OSG_FN(osg_attention) OSG_FN(osg_attention_strided) OSG_FN(osg_instance_norm)
    OSG_FN(osg_group_norm_nhwc) OSG_FN(osg_group_norm_conv3x3_supported) OSG_FN(osg_group_norm_conv3x3) OSG_FN(osg_layer_norm) OSG_FN(osg_reduce_mean_last) OSG_FN(osg_softmax_last) OSG_FN(osg_unary)
    OSG_FN(osg_binary) OSG_FN(osg_geglu) OSG_FN(osg_transpose) OSG_FN(osg_copy_2d) OSG_FN(osg_concat2) OSG_FN(osg_resize_nearest)
    OSG_FN(osg_gather_rows) OSG_FN(osg_maxpool_nhwc) OSG_FN(osg_convert) OSG_FN(osg_sampler_prepare) OSG_FN(osg_sampler_cfg_euler_a)
#undef OSG_FN
};

class HipBackend {
public:
    OsgApi api;
    osg_ctx* ctx = nullptr;

    explicit HipBackend(int device) {
        std::string path;
        if (const char* e = std::getenv("OSGPU_LIB")) path = e;
        if (path.empty()) {
            Dl_info info;
            if (dladdr((void*)&HipBackend::anchor, &info) && info.dli_fname) {
                std::string self = info.dli_fname;
                auto p = self.find_last_of('/');
                path = (p == std::string::npos ? std::string(".") : self.substr(0, p)) + "/libosgpu.so";
            } else {
                path = "libosgpu.so";
            }
        }
        m_handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!m_handle) throw std::runtime_error("HipBackend: unable to load " + path + ": " + dlerror() + " (no CPU fallback exists)");
#define OSG_FN(name)                                                                                   \
    api.name = reinterpret_cast<decltype(api.name)>(dlsym(m_handle, #name));                           \
    if (!api.name) throw std::runtime_error(std::string("HipBackend: symbol missing in libosgpu.so: ") + #name);
        OSG_FN(osg_device_count) OSG_FN(osg_init) OSG_FN(osg_destroy) OSG_FN(osg_last_error) OSG_FN(osg_device_name) OSG_FN(osg_stream) OSG_FN(osg_set_autotune)
        OSG_FN(osg_malloc) OSG_FN(osg_free) OSG_FN(osg_upload) OSG_FN(osg_upload_sync) OSG_FN(osg_host_register) OSG_FN(osg_host_unregister) OSG_FN(osg_upload_pinned) OSG_FN(osg_download) OSG_FN(osg_copy)
        OSG_FN(osg_memset) OSG_FN(osg_sync) OSG_FN(osg_graph_begin) OSG_FN(osg_graph_end) OSG_FN(osg_graph_launch)
        OSG_FN(osg_graph_destroy) OSG_FN(osg_side_begin) OSG_FN(osg_side_end) OSG_FN(osg_side_join) OSG_FN(osg_timer_start) OSG_FN(osg_timer_stop) OSG_FN(osg_conv2d_nhwc) OSG_FN(osg_conv2d_nhwc_rb) OSG_FN(osg_gemm) OSG_FN(osg_gemm_ln) OSG_FN(osg_gemm_rowstats) OSG_FN(osg_gemm_w8) OSG_FN(osg_conv2d_nhwc_w8)
        OSG_FN(osg_transpose_kn_to_nk) OSG_FN(osg_attention) OSG_FN(osg_attention_strided) OSG_FN(osg_instance_norm)
        OSG_FN(osg_group_norm_nhwc) OSG_FN(osg_group_norm_conv3x3_supported) OSG_FN(osg_group_norm_conv3x3) OSG_FN(osg_layer_norm) OSG_FN(osg_reduce_mean_last) OSG_FN(osg_softmax_last) OSG_FN(osg_unary)
        OSG_FN(osg_binary) OSG_FN(osg_geglu) OSG_FN(osg_transpose) OSG_FN(osg_copy_2d) OSG_FN(osg_concat2) OSG_FN(osg_resize_nearest)
        OSG_FN(osg_gather_rows) OSG_FN(osg_maxpool_nhwc) OSG_FN(osg_convert) OSG_FN(osg_sampler_prepare) OSG_FN(osg_sampler_cfg_euler_a)
#undef OSG_FN
        if (api.osg_device_count() <= 0)
            throw std::runtime_error("HipBackend: no HIP device visible; the MI355X backend has no CPU fallback");
        int rc = api.osg_init(device, &ctx);
        if (rc) throw std::runtime_error("HipBackend: osg_init(device " + std::to_string(device) + ") failed with code " + std::to_string(rc));
    }

    ~HipBackend() {
        if (ctx) api.osg_destroy(ctx);
        // the library stays loaded: kernels' code objects must outlive any late stream callbacks
    }

    HipBackend(const HipBackend&) = delete;
    HipBackend& operator=(const HipBackend&) = delete;

    // throws the reference-style std::runtime_error carrying the backend's message
    void check(int rc, const char* what) const {
        if (rc) throw std::runtime_error(std::string(what) + ": " + api.osg_last_error(ctx));
    }

    void* malloc(size_t bytes) {
        void* p = nullptr;
        check(api.osg_malloc(ctx, bytes, &p), "osg_malloc");
        return p;
    }
    void free(void* p) { if (p) api.osg_free(ctx, p); }

private:
    static void anchor() {}
    void* m_handle = nullptr;
};

}  // namespace onnxstream
