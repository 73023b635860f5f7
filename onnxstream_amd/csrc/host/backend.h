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

// every entry point of include/osgpu.h the host binds (one list: declaration of the table and its dlsym loop)
#define OSG_API_LIST(X) \
    X(osg_device_count) X(osg_init) X(osg_destroy) X(osg_last_error) X(osg_device_name) X(osg_stream) \
    X(osg_set_autotune) X(osg_malloc) X(osg_free) X(osg_upload) X(osg_upload_sync) X(osg_host_register) \
    X(osg_host_unregister) X(osg_upload_pinned) X(osg_upload_pinned_async) X(osg_copy_fence) X(osg_download) X(osg_copy) X(osg_memset) X(osg_sync) \
    X(osg_graph_begin) X(osg_graph_end) X(osg_graph_launch) X(osg_graph_destroy) \
    X(osg_timer_start) X(osg_timer_stop) X(osg_conv2d_nhwc) X(osg_conv2d_nhwc_rb) X(osg_conv2d_nhwc_v) X(osg_set_stat_sinks) X(osg_group_norm_stats_nhwc) X(osg_gemm) \
    X(osg_gemm_ln) X(osg_gemm_rowstats) X(osg_gemm_w8) X(osg_conv2d_nhwc_w8) X(osg_gemm_w8_v) X(osg_conv2d_nhwc_w8_v) X(osg_transpose_kn_to_nk) X(osg_attention) \
    X(osg_attention_strided) X(osg_sdpa) X(osg_rms_norm) X(osg_rope) X(osg_instance_norm) X(osg_group_norm_nhwc) X(osg_layer_norm) \
    X(osg_reduce_mean_last) X(osg_softmax_last) X(osg_unary) X(osg_binary) X(osg_geglu) X(osg_transpose) \
    X(osg_copy_2d) X(osg_concat2) X(osg_resize_nearest) X(osg_gather_rows) X(osg_maxpool_nhwc) X(osg_convert) \
    X(osg_sampler_prepare) X(osg_sampler_cfg_euler_a) X(osg_qu8_conv2d_nhwc) X(osg_qu8_conv2d_nhwc_t) X(osg_qu8_conv_tap_sums) X(osg_qu8_gemm) X(osg_qu8_lut) X(osg_qu8_binary) \
    X(osg_qu8_instance_norm) X(osg_qu8_instance_norm_nhwc) X(osg_qu8_affine_act) X(osg_qu8_norm_affine_act_nhwc) X(osg_qu8_softmax_last) X(osg_range_push) X(osg_range_pop) X(osg_marker_record) \
    X(osg_copy_wait_marker) X(osg_timer_mark) X(osg_timer_between) X(osg_tblock_tail_supported) X(osg_tblock_tail) X(osg_tblock_kv_pack_elems) X(osg_tblock_kv_pack_jobs) X(osg_tblock_pack_weight)

struct OsgApi {
#define OSG_FN(name) decltype(&::name) name = nullptr;
    OSG_API_LIST(OSG_FN)
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
        OSG_API_LIST(OSG_FN)
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
