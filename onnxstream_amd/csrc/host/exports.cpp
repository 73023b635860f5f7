// exports.cpp -- the flat C API over Model that foreign-language bindings use.  Names, argument order, ownership and
// error conventions are those of the reference's src/exports.cpp (model_new :42, model_new_2 :62, model_delete :87,
// model_read_string :92, model_read_file :98, model_get_weights_names :111, model_add_weights_file :150,
// model_add_tensor :169, model_get_tensor :205, model_get_all_tensor_names :235, model_run :245, model_run_2 :258,
// model_clear_tensors :271, model_set_option :276, model_add_extra_output :303, model_free_buffer :308): errors come back
// as malloc'd C strings the caller releases with model_free_buffer.  Extra entry points for this backend carry a
// `model_hip_` prefix.
#include <cstdio>
#include <cstring>

#include "onnxstream.h"

using namespace onnxstream;

namespace {

char* dup_cstr(const std::string& s) {
    char* p = (char*)std::malloc(s.size() + 1);
    std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

struct Handle {
    explicit Handle(int threads) : model(threads) {}
    Model model;
    std::string definition;
    std::string provider;
};

const char* dtype_name(TensorDataType t) {
    switch (t) {
        case TensorDataType::uint8: return "uint8";
        case TensorDataType::float16: return "float16";
        case TensorDataType::float32: return "float32";
        case TensorDataType::int64: return "int64";
        default: throw std::invalid_argument("Unsupported tensor data format.");
    }
}

}  // namespace

extern "C" {

Handle* model_new_2(int threads_count, char* wp_name) {
    Handle* h = new Handle(threads_count);
    h->provider = wp_name;
    const std::string& w = h->provider;
    if (w == "ram") h->model.set_weights_provider(RamWeightsProvider<WeightsProvider>());
    else if (w == "nocache") h->model.set_weights_provider(DiskNoCacheWeightsProvider());
    else if (w == "prefetch") h->model.set_weights_provider(DiskPrefetchWeightsProvider());
    else if (w == "ram+nocache") h->model.set_weights_provider(RamWeightsProvider<DiskNoCacheWeightsProvider>(DiskNoCacheWeightsProvider()));
    else if (w == "ram+prefetch") h->model.set_weights_provider(RamWeightsProvider<DiskPrefetchWeightsProvider>(DiskPrefetchWeightsProvider()));
    else {
        delete h;
        return nullptr;
    }
    return h;
}

Handle* model_new() {
    static char ram[] = "ram";
    return model_new_2(0, ram);
}

void model_delete(Handle* h) { delete h; }

void model_read_string(Handle* h, char* str) {
    h->definition = str;
    h->model.read_string(str);
}

char* model_read_file(Handle* h, char* fn) {
    try {
        h->model.read_file(fn);
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}

char* model_get_weights_names(Handle* h) {
    Model probe(-1);
    probe.m_support_dynamic_shapes = true;
    probe.set_weights_provider(CollectNamesWeightsProvider(true));
    probe.read_string(h->definition.c_str());
    probe.init();
    std::string out;
    for (auto& e : probe.get_weights_provider<CollectNamesWeightsProvider>().m_names_vec) {
        std::string fn = e.m_name;
        auto pos = fn.find("_nchw.bin");
        if (pos != std::string::npos) fn = fn.substr(0, pos) + "_nhwc.bin";
        if (!out.empty()) out += "|";
        out += std::string(dtype_name(e.m_type)) + ":" + fn;
    }
    return dup_cstr(out);
}

// 64-bit size variant (bindings that pass size_t need not truncate: a buffer of 4 GiB or more handed to the 32-bit entry point would
// be allocated short and overrun by the caller); the reference's 32-bit entry point (src/exports.cpp:139) forwards to it
void* model_hip_add_weights_file(Handle* h, const char* type, const char* name, unsigned long long size) {
    if (h->provider != "ram") return nullptr;
    auto& wp = h->model.get_weights_provider<RamWeightsProvider<WeightsProvider>>();
    const std::string t = type;
    if (t == "uint8") return wp.add_empty_and_return_ptr<uint8_t>(name, (size_t)size / sizeof(uint8_t));
    if (t == "float16") return wp.add_empty_and_return_ptr<uint16_t>(name, (size_t)size / sizeof(uint16_t));
    if (t == "float32") return wp.add_empty_and_return_ptr<float>(name, (size_t)size / sizeof(float));
    if (t == "int64") return wp.add_empty_and_return_ptr<int64_t>(name, (size_t)size / sizeof(int64_t));
    throw std::invalid_argument("Unsupported tensor data format.");
}

void* model_add_weights_file(Handle* h, char* type, char* name, unsigned int size) {
    return model_hip_add_weights_file(h, type, name, (unsigned long long)size);
}

void* model_add_tensor(Handle* h, char* type, char* name, unsigned int dims_num, unsigned int* dims) {
    Tensor t;
    t.m_name = name;
    size_t n = 1;
    for (unsigned i = 0; i < dims_num; i++) {
        t.m_shape.push_back(dims[i]);
        n *= dims[i];
    }
    const std::string ty = type;
    void* ptr = nullptr;
    if (ty == "float32") {
        tensor_vector<float> v(n);
        ptr = v.data();  // the heap block moves with the vector into the tensor
        t.set_vector(std::move(v));
    } else if (ty == "int64") {
        tensor_vector<int64_t> v(n);
        ptr = v.data();
        t.set_vector(std::move(v));
    } else {
        throw std::invalid_argument("Unsupported tensor data format.");
    }
    h->model.push_tensor(std::move(t));
    return ptr;
}

void* model_get_tensor(Handle* h, char* name) {
    struct Ret {
        size_t dims_num;
        size_t* dims;
        size_t data_num;
        float* data;
    };
    for (auto& t : h->model.m_data)
        if (t.m_name == name) {
            if (t.m_type != TensorDataType::float32) return nullptr;
            Ret* r = (Ret*)std::malloc(sizeof(Ret));
            r->dims_num = t.m_shape.size();
            r->dims = t.m_shape.data();
            r->data_num = t.get_vector<float>().size();
            r->data = t.get_vector<float>().data();
            return r;
        }
    return nullptr;
}

// sample `index` of a tensor that was produced for several pushed samples (index 0 == model_get_tensor); the reference app reaches
// the same data through Tensor::m_batch (src/sd.cpp:1098-1161 SDCoroState::get_result)
void* model_hip_get_tensor_batch(Handle* h, char* name, unsigned int index) {
    struct Ret {
        size_t dims_num;
        size_t* dims;
        size_t data_num;
        float* data;
    };
    for (auto& t : h->model.m_data)
        if (t.m_name == name) {
            Tensor* src = &t;
            if (index > 0) {
                if (!t.m_batch || index - 1 >= t.m_batch->size()) return nullptr;
                src = &(*t.m_batch)[index - 1];
            }
            if (src->m_type != TensorDataType::float32) return nullptr;
            Ret* r = (Ret*)std::malloc(sizeof(Ret));
            r->dims_num = src->m_shape.size();
            r->dims = src->m_shape.data();
            r->data_num = src->get_vector<float>().size();
            r->data = src->get_vector<float>().data();
            return r;
        }
    return nullptr;
}

char* model_get_all_tensor_names(Handle* h) {
    std::string out;
    for (auto& t : h->model.m_data) out += (out.empty() ? "" : "|") + t.m_name;
    return dup_cstr(out);
}

void model_run(Handle* h) {
    try {
        h->model.run();
    } catch (const std::exception& e) {
        printf("=== ERROR === %s\n", e.what());
        throw;
    }
}

char* model_run_2(Handle* h) {
    try {
        h->model.run();
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}

void model_clear_tensors(Handle* h) { h->model.m_data.clear(); }

void model_set_option(Handle* h, char* name, unsigned int value) {
    Model& m = h->model;
    const std::string n = name;
    const bool b = value != 0;
    if (n == "use_fp16_arithmetic") m.m_use_fp16_arithmetic = b;
    else if (n == "use_uint8_qdq") m.m_use_uint8_qdq = b;
    else if (n == "use_uint8_arithmetic") m.m_use_uint8_arithmetic = b;
    else if (n == "fuse_ops_in_attention") m.m_fuse_ops_in_attention = b;
    else if (n == "force_fp16_storage") m.m_force_fp16_storage = b;
    else if (n == "support_dynamic_shapes") m.m_support_dynamic_shapes = b;
    else if (n == "use_ops_cache") m.m_use_ops_cache = b;
    else if (n == "use_scaled_dp_attn_op") m.m_use_scaled_dp_attn_op = b;
    else if (n == "use_next_op_cache") m.m_use_next_op_cache = b;
    else if (n == "ops_printf") m.m_ops_printf = b;
    else if (n == "ops_times_printf") m.m_ops_times_printf = b;
    else if (n == "use_nchw_convs") m.m_use_nchw_convs = b;
    else if (n == "range_data_calibrate") m.m_range_data_calibrate = b;
    // ---- backend additions ----
    else if (n == "attention_fused_ops_parts") m.m_attention_fused_ops_parts = value;
    else if (n == "hip_device") m.m_hip_device = (int)value;
    else if (n == "hip_fusion_level") m.m_hip_fusion_level = (int)value;
    else if (n == "hip_use_graph") m.m_hip_use_graph = b;
    else if (n == "hip_autotune") m.m_hip_autotune = b;
    else if (n == "hip_fuse_ln_gemm") m.m_hip_fuse_ln_gemm = b;
    else if (n == "hip_concat_views") m.m_hip_concat_views = b;
    else if (n == "hip_fuse_tblock") m.m_hip_fuse_tblock = b;
    else if (n == "hip_gn_stats") m.m_hip_gn_stats = (int)value;
    else if (n == "hip_stream_weights") m.m_hip_stream_weights = b;
    else if (n == "hip_w8_resident") m.m_hip_w8_resident = b;
    else if (n == "hip_resident_outputs") m.m_hip_resident_outputs = b;
    else {
        const char* err = "model_set_option: 'name' not found.";
        printf("=== ERROR === %s\n", err);
        throw std::invalid_argument(err);
    }
}

void model_add_extra_output(Handle* h, char* name) { h->model.m_extra_outputs.emplace_back(name); }

void model_free_buffer(void* ptr) { std::free(ptr); }

// ---- backend additions -------------------------------------------------------------------------------------------
double model_hip_last_pass_ms(Handle* h) { return h->model.hip_last_pass_ms(); }
unsigned long long model_hip_last_kernel_count(Handle* h) { return h->model.hip_last_kernel_count(); }
void model_hip_invalidate_plan(Handle* h) { h->model.hip_invalidate_plan(); }
unsigned long long model_hip_plans_built(Handle* h) { return h->model.hip_plans_built(); }
// bytes the last pass pulled through the WeightsProvider and streamed host->device (0 in resident mode)
unsigned long long model_hip_streamed_bytes(Handle* h) { return h->model.hip_streamed_bytes(); }
unsigned long long model_hip_resident_weight_bytes(Handle* h) { return h->model.hip_resident_weight_bytes(); }
// CudaOptions::m_vram_to_use through the C API (the reference sets it from C++ only: src/llm.cpp set_cuda_options)
void model_hip_set_vram_budget(Handle* h, unsigned long long bytes) { h->model.set_cuda_options(CudaOptions(bytes, false)); }
// Model::m_requires_upcast is a std::function (set from C++ by src/llm.cpp:379-383: ops whose NAME contains "/input_layernorm/" or
// "/post_attention_layernorm/" run in fp32): through the C API as a '|'-separated list of substrings of the op name; "" clears it
void model_hip_set_upcast_substrings(Handle* h, const char* list) {
    std::vector<std::string> subs;
    std::string cur;
    for (const char* p = list ? list : ""; ; p++) {
        if (*p == '|' || *p == 0) {
            if (!cur.empty()) subs.push_back(cur);
            cur.clear();
            if (!*p) break;
        } else cur.push_back(*p);
    }
    if (subs.empty()) h->model.m_requires_upcast = nullptr;
    else
        h->model.m_requires_upcast = [subs](const std::string&, const std::string& name) {
            for (auto& s : subs)
                if (name.find(s) != std::string::npos) return true;
            return false;
        };
    h->model.hip_invalidate_plan();
}
// What src/llm.cpp does to Model::m_data and m_outputs_convert_set from C++ between two calls (:366, :403-407): keep an output out of the
// fp32 conversion, and hand an output of the last run back as an input of the next one under another name
void model_hip_add_outputs_convert(Handle* h, char* name) { h->model.m_outputs_convert_set.insert(name); }
int model_hip_drop_tensor(Handle* h, char* name) {   // (llm.cpp's get_output moves a result out of m_data, :342-353)
    auto& d = h->model.m_data;
    for (size_t i = 0; i < d.size(); i++)
        if (d[i].m_name == name) { d.erase(d.begin() + i); return 1; }
    return 0;
}
// bring a device-resident tensor (hip_resident_outputs) to the host; returns an error string or NULL
char* model_hip_fetch_tensor(Handle* h, char* name) {
    try {
        h->model.hip_fetch_tensor(name);
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}
int model_hip_rename_tensor(Handle* h, char* from, char* to) {
    for (auto& t : h->model.m_data)
        if (t.m_name == from) { t.m_name = to; return 1; }
    return 0;
}
// relaunch the captured pass n times on the resident inputs; ms_each (may be NULL) receives per-launch device times
char* model_hip_replay(Handle* h, int n, float* ms_each) {
    try {
        h->model.hip_replay(n, ms_each);
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}
// refresh sample `index` of a resident graph input (fp32, `count` elements) without running a pass
char* model_hip_set_input(Handle* h, char* name, long long index, const float* data, unsigned long long count) {
    try {
        h->model.hip_set_input(name, (long)index, data, (size_t)count);
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}
// the reference app's denoising loop (CFG combine + Euler-Ancestral, src/sd.cpp:1397-1559, src/samplers.h:1430-1472) enqueued on the device
// without per-step host round trips.  x:[prompts,L] fp32 is updated in place; noise:[steps,prompts,L]; the five per-step scalar arrays
// have `steps` entries; clip (may be NULL) is a sixth: per-step clamp of the new latents, 0 = none.  *ms (may be NULL) receives the device time of the loop.
char* model_hip_sampler_loop(Handle* h, char* sample_name, char* timestep_name, char* out_name, int steps, int prompts, float* x, const float* noise,
                             const float* c_in, const float* c_out, const float* t, const float* sigma, const float* d_sigma, const float* sigma_up, float guidance, const float* clip,
                             double* ms) {
    try {
        const double v = h->model.hip_sampler_loop(sample_name, timestep_name, out_name, steps, prompts, x, noise, c_in, c_out, t, sigma, d_sigma, sigma_up, guidance, clip);
        if (ms) *ms = v;
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}
// steps and arena placement of the current plan (malloc'ed text, free with model_free_buffer); "ERROR: ..." on error
char* model_hip_plan_info(Handle* h) {
    try {
        return dup_cstr(h->model.hip_plan_info());
    } catch (const std::exception& e) {
        return dup_cstr(std::string("ERROR: ") + e.what());
    }
}
// per-step timing report (malloc'ed text, free with model_free_buffer); on error the text starts with "ERROR: "
char* model_hip_profile(Handle* h, int reps) {
    try {
        return dup_cstr(h->model.hip_profile(reps));
    } catch (const std::exception& e) {
        return dup_cstr(std::string("ERROR: ") + e.what());
    }
}

// range_data.txt (per-op output ranges of a calibration run; src/sd.cpp:1214 reads it before the uint8 VAE decode, :1241 writes it after
// --decoder-calibrate): Model::read_range_data / write_range_data have no export in the reference's C API (the app links the class)
char* model_hip_read_range_data(Handle* h, const char* filename) {
    try {
        h->model.read_range_data(filename);
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}
char* model_hip_write_range_data(Handle* h, const char* filename) {
    try {
        h->model.write_range_data(filename);
        return nullptr;
    } catch (const std::exception& e) {
        return dup_cstr(e.what());
    }
}

}  // extern "C"
