// TEST INFRASTRUCTURE ONLY (oracle). Never linked into the product libraries.
//
// ref_extra.cpp -- a few extra extern "C" accessors on the *reference* onnxstream::Model for the
// public fields that /root/reference/src/exports.cpp does not expose through model_set_option
// (attention parts, range data, calibration).  Compiled together with the unmodified reference
// sources into oracle/_ref/libonnxstream_ref.so; it includes the reference header from where it lies.
//
// The handle is the ModelContext* returned by the reference's own model_new_2 (exports.cpp:62);
// its first member is the Model (exports.cpp:28-40), so the handle doubles as a Model*.
#include "onnxstream.h"
#include <cstring>

using namespace onnxstream;

static Model* as_model(void* ctx) { return reinterpret_cast<Model*>(ctx); }

extern "C" {

void ref_set_attention_parts(void* ctx, unsigned parts) { as_model(ctx)->m_attention_fused_ops_parts = parts; }

void ref_set_range_data_calibrate(void* ctx, unsigned on) { as_model(ctx)->m_range_data_calibrate = on != 0; }

const char* ref_read_range_data(void* ctx, const char* fn) {
    static thread_local std::string err;
    try { as_model(ctx)->read_range_data(fn); return nullptr; }
    catch (const std::exception& e) { err = e.what(); return err.c_str(); }
}

const char* ref_write_range_data(void* ctx, const char* fn) {
    static thread_local std::string err;
    try { as_model(ctx)->write_range_data(fn); return nullptr; }
    catch (const std::exception& e) { err = e.what(); return err.c_str(); }
}

// Model::m_requires_upcast (a std::function the LLM app sets from C++, src/llm.cpp:379-383) as a '|'-separated list of op-name substrings
void ref_set_upcast_substrings(void* ctx, const char* list) {
    std::vector<std::string> subs;
    std::string cur;
    for (const char* p = list ? list : ""; ; p++) {
        if (*p == '|' || *p == 0) {
            if (!cur.empty()) subs.push_back(cur);
            cur.clear();
            if (!*p) break;
        } else cur.push_back(*p);
    }
    if (subs.empty()) as_model(ctx)->m_requires_upcast = nullptr;
    else
        as_model(ctx)->m_requires_upcast = [subs](const std::string&, const std::string& name) {
            for (auto& s : subs)
                if (name.find(s) != std::string::npos) return true;
            return false;
        };
}

int ref_drop_tensor(void* ctx, const char* name) {
    auto& d = as_model(ctx)->m_data;
    for (size_t i = 0; i < d.size(); i++)
        if (d[i].m_name == name) { d.erase(d.begin() + i); return 1; }
    return 0;
}
int ref_rename_tensor(void* ctx, const char* from, const char* to) {
    for (auto& t : as_model(ctx)->m_data)
        if (t.m_name == from) { t.m_name = to; return 1; }
    return 0;
}
void ref_add_outputs_convert_exclusion(void* ctx, const char* name) { as_model(ctx)->m_outputs_convert_set.insert(name); }

void ref_add_force_uint8_storage(void* ctx, const char* name) { as_model(ctx)->m_force_uint8_storage_set.insert(name); }

// Generic tensor read-back (any dtype).  Returns element count, fills dtype (1=u8, 2=f16, 3=f32, 4=i64),
// rank/shape (up to 8 dims), scale/zero-point and the raw data pointer (owned by the model).
size_t ref_get_tensor_any(void* ctx, const char* name, int* dtype, size_t* rank, size_t* shape, float* scale, int* zp,
                          const void** data) {
    for (auto& t : as_model(ctx)->m_data)
        if (t.m_name == name) {
            *dtype = (int)t.m_type;
            *rank = t.m_shape.size();
            for (size_t i = 0; i < t.m_shape.size() && i < 8; i++) shape[i] = t.m_shape[i];
            *scale = t.m_scale;
            *zp = t.m_zero_point;
            switch (t.m_type) {
                case TensorDataType::uint8: *data = t.get_vector<uint8_t>().data(); return t.get_vector<uint8_t>().size();
                case TensorDataType::float16: *data = t.get_vector<uint16_t>().data(); return t.get_vector<uint16_t>().size();
                case TensorDataType::float32: *data = t.get_vector<float>().data(); return t.get_vector<float>().size();
                case TensorDataType::int64: *data = t.get_vector<int64_t>().data(); return t.get_vector<int64_t>().size();
                default: return 0;
            }
        }
    return 0;
}

// Push a filled fp32 tensor through Model::push_tensor, as the reference app does for the VAE latents (src/sd.cpp:1228-1236): with
// m_use_uint8_arithmetic set, push_tensor quantises the REAL data (model_add_tensor pushes an empty tensor first and cannot be used).
const char* ref_push_tensor_f32(void* ctx, const char* name, unsigned dims_num, const unsigned* dims, const float* data) {
    static thread_local std::string err;
    try {
        onnxstream::Tensor t;
        t.m_name = name;
        size_t n = 1;
        for (unsigned i = 0; i < dims_num; i++) { t.m_shape.push_back(dims[i]); n *= dims[i]; }
        onnxstream::tensor_vector<float> v(data, data + n);
        t.set_vector(std::move(v));
        as_model(ctx)->push_tensor(std::move(t));
        return nullptr;
    } catch (const std::exception& e) { err = e.what(); return err.c_str(); }
}

}  // extern "C"
