// plan.cpp -- lowering of an OnnxStream graph onto the HIP operator layer (see plan.h for the pipeline).
// Per-op semantics follow the reference's Model::run branches (cited per function as reference src/onnxstream.cpp:LINE);
// the error strings mirror the reference's so that callers matching on them keep working.
#include "plan.h"
#include "qu8.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <string_view>
#include <unordered_map>
#include <unordered_set>

namespace onnxstream {

namespace {

using Shape = std::vector<long>;

long prod(const Shape& s, size_t from = 0, size_t to = (size_t)-1) {
    long n = 1;
    for (size_t i = from; i < std::min(to, s.size()); i++) n *= s[i];
    return n;
}

Shape to_shape(const std::vector<size_t>& s) { return Shape(s.begin(), s.end()); }

std::string shape_str(const Shape& s) {
    std::string r = "(";
    for (size_t i = 0; i < s.size(); i++) r += (i ? "," : "") + std::to_string(s[i]);
    return r + ")";
}

std::vector<int> int_list(const std::string& s) {
    std::vector<int> out;
    size_t b = 0;
    while (b <= s.size()) {
        size_t e = s.find(',', b);
        if (e == std::string::npos) e = s.size();
        if (e > b) out.push_back(std::stoi(s.substr(b, e - b)));
        b = e + 1;
    }
    return out;
}

const std::string* attr(const Operation& op, const char* key) {
    for (auto& a : op.m_attributes)
        if (a.first == key) return &a.second;
    return nullptr;
}

size_t esize(osg_dtype d) { return d == OSG_U8 ? 1 : d == OSG_F16 ? 2 : d == OSG_F32 ? 4 : 8; }

float half_to_float(uint16_t h) {
    uint32_t sign = (h >> 15) & 1, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    float v;
    if (exp == 0) v = std::ldexp((float)man, -24);
    else if (exp == 31) v = man ? NAN : INFINITY;
    else v = std::ldexp((float)(man | 0x400), (int)exp - 25);
    return sign ? -v : v;
}

uint16_t float_to_half(float f) {   // round-to-nearest-even, like the reference's fp32 -> fp16 convert
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

bool is_const_tensor(const Tensor& t) { return !t.m_name.empty() && t.m_type != TensorDataType::none; }

}  // namespace

// ======================================================================================================================
Plan::Plan(Model& model, HipBackend& backend, ConstPool& cpool, size_t batch) : m(model), be(backend), pool(cpool), N((long)batch) {
    fp16 = m.m_use_fp16_arithmetic;
    fusion = fusion_req = m.m_hip_fusion_level;
    stream_weights = m.m_hip_stream_weights;
    // CudaOptions::m_vram_to_use (reference :396-398: weights are placed on the GPU until the budget is spent, the rest stay off it): here the
    // weights beyond the budget are streamed every pass through a device ring instead -- a budget switches the streamed-weights mode on
    vram_budget = (size_t)m.m_cuda_options.m_vram_to_use;
    budgeted = vram_budget > 0;
    if (budgeted) stream_weights = true;
    w8_resident = m.m_hip_w8_resident && !stream_weights;
    fuse_ln_gemm = m.m_hip_fuse_ln_gemm;
    concat_views = m.m_hip_concat_views;
    fuse_tblock = m.m_hip_fuse_tblock;
    gn_stats_req = m.m_hip_gn_stats;
    gn_stats_min_elems = m.m_hip_gn_stats == 2 ? (8L << 20) : 0;
    gn_stats_on = m.m_hip_gn_stats != 0 && !stream_weights && m.m_hip_fusion_level >= 2 && !m.m_use_uint8_arithmetic && !m.m_range_data_calibrate;
    u8 = m.m_use_uint8_arithmetic;
    u8_qdq = m.m_use_uint8_qdq;
    autotune = m.m_hip_autotune;
    calibrate = m.m_range_data_calibrate;
    fuse_attn = m.m_fuse_ops_in_attention;
    sdp_attn = m.m_use_scaled_dp_attn_op;
    outputs_convert_set = m.m_outputs_convert_set;
    extra_outputs = m.m_extra_outputs;
    recycle = m.m_support_dynamic_shapes && !stream_weights;
    resident_outputs = m.m_hip_resident_outputs && m.m_support_dynamic_shapes && !stream_weights && !m.m_outputs_convert_set.empty();
}

bool Plan::compatible(Model& mm, size_t batch) const {
    static const bool say = getenv("OSG_PLAN_TIMING") != nullptr;       // (with the other plan diagnostics: WHY a call re-plans)
    auto no = [&](const char* why) {
        if (say) fprintf(stderr, "[plan] re-plan: %s\n", why);
        return false;
    };
    const bool want_stream = mm.m_hip_stream_weights || mm.m_cuda_options.m_vram_to_use > 0;
    if ((long)batch != N) return no("batch size");
    if (mm.m_use_fp16_arithmetic != fp16 || mm.m_use_uint8_arithmetic != u8 || mm.m_use_uint8_qdq != u8_qdq) return no("arithmetic type");
    if (mm.m_hip_fusion_level != fusion_req || mm.m_hip_fuse_ln_gemm != fuse_ln_gemm || mm.m_hip_concat_views != concat_views || mm.m_hip_fuse_tblock != fuse_tblock || mm.m_hip_gn_stats != gn_stats_req ||
        mm.m_fuse_ops_in_attention != fuse_attn || mm.m_use_scaled_dp_attn_op != sdp_attn || mm.m_hip_autotune != autotune)
        return no("fusion / tuning options");
    if (want_stream != stream_weights || (size_t)mm.m_cuda_options.m_vram_to_use != vram_budget ||
        (mm.m_hip_w8_resident && !want_stream) != w8_resident)
        return no("weight residency options");
    if (mm.m_extra_outputs != extra_outputs || mm.m_outputs_convert_set != outputs_convert_set ||
        (mm.m_hip_resident_outputs && mm.m_support_dynamic_shapes && !want_stream && !mm.m_outputs_convert_set.empty()) != resident_outputs)
        return no("output set");
    if (mm.m_range_data_calibrate != calibrate) return no("calibration mode");
    // a pushed input with another shape / type (dynamic-shape models) re-plans, the way the reference simply re-executes (:3550)
    for (auto& in : inputs)
        for (auto& t : mm.m_data)
            if (t.m_name == in.name) {
                if (t.m_type != in.host_type || t.m_shape != in.shape) return no("an input's type or shape");
                if (t.m_hip_resident != in.resident) return no("a device-resident input");   // read at ITS address: another buffer (or a host tensor) re-plans
                if (t.m_type == TensorDataType::int64) {
                    auto& v = t.get_vector<int64_t>();
                    if (v.size() != in.ivals.size() || !std::equal(v.begin(), v.end(), in.ivals.begin())) return no("an int64 input's values");
                }
            }
    return true;
}

int Plan::new_val(const std::string& name, const Shape& shape, osg_dtype dt, Lay lay, bool batched) {
    Val v;
    v.name = name;
    v.shape = shape;
    v.dtype = dt;
    v.lay = lay;
    v.batched = batched;
    vals.push_back(std::move(v));
    int id = (int)vals.size() - 1;
    if (!name.empty()) by_name[name] = id;
    return id;
}

int Plan::root_of(int v) const {
    while (vals[v].root >= 0) v = vals[v].root;
    return v;
}

int Plan::alias(int v, const Shape& shape, Lay lay, const std::string& name) {
    Val a;
    a.name = name;
    a.shape = shape;
    a.dtype = vals[v].dtype;
    a.lay = lay;
    a.batched = vals[v].batched;
    a.is_const = vals[v].is_const;
    a.root = v;
    a.qscale = vals[v].qscale;
    a.qzp = vals[v].qzp;
    a.qdyn = vals[v].qdyn;
    a.host_f = vals[v].host_f;
    a.host_i = vals[v].host_i;
    a.host_valid = vals[v].host_valid;
    vals.push_back(std::move(a));
    int id = (int)vals.size() - 1;
    if (!name.empty()) by_name[name] = id;
    return id;
}

const Val& Plan::qv(int v) const {
    for (int guard = 0; guard < 64; guard++) {
        if (vals[v].qsrc >= 0) v = vals[v].qsrc;
        else if (vals[v].root >= 0) v = vals[v].root;
        else break;
    }
    return vals[v];
}

void Plan::share_q(int dst, int src) {
    if (vals[src].dtype != OSG_U8 || dst == src) return;
    int s = src;
    while (vals[s].qsrc < 0 && vals[s].root >= 0) s = vals[s].root;
    if (vals[s].qsrc >= 0) s = vals[s].qsrc;
    int d = dst;
    while (vals[d].root >= 0) d = vals[d].root;
    if (d == s) return;
    vals[d].qsrc = s;
    vals[d].qscale = vals[s].qscale;
    vals[d].qzp = vals[s].qzp;
    vals[d].qdyn = vals[s].qdyn;
}

long Plan::total_elems(int v) const { return vals[v].numel() * (vals[v].batched ? N : 1); }
size_t Plan::val_bytes(int v) const { return (size_t)total_elems(v) * esize(vals[v].dtype); }

void* Plan::ptr(int v) const {
    size_t off = 0;
    while (vals[v].root >= 0) {
        off += vals[v].view_off;
        v = vals[v].root;
    }
    const Val& r = vals[v];
    if (r.dptr) return (char*)r.dptr + off;
    return (char*)arena + r.offset + off;
}

bool Plan::in_one_slab(const char* lo, const char* hi) const {
    for (auto& sl : slabs)
        if (lo >= sl.first && hi <= sl.first + sl.second) return true;
    return false;
}

void* Plan::small_alloc(size_t bytes) {
    const size_t need = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    if (need > ((size_t)1 << 20)) {
        void* p = be.malloc(need);
        owned.push_back(p);
        return p;
    }
    if (need > slab_left) {
        slab_left = (size_t)8 << 20;
        slab = (char*)pooled_malloc(slab_left);
        slabs.push_back({slab, slab_left});
    }
    void* p = slab;
    slab += need;
    slab_left -= need;
    return p;
}

// a buffer that returns to the Model's pool when this plan goes (models that re-plan on every call), else an owned allocation
void* Plan::pooled_malloc(size_t bytes) {
    if (!recycle) {
        void* p = be.malloc(bytes);
        owned.push_back(p);
        return p;
    }
    size_t have = bytes;   // (a spare may be larger than asked for: it goes back with its real size)
    void* p = pool.take(bytes, &have);
    if (!p) p = be.malloc(bytes);
    recyclable.push_back({p, have});
    return p;
}

void* Plan::const_alloc(const std::string& tag, size_t bytes, bool* fresh) {
    if (stream_weights || tag.empty()) {
        *fresh = true;
        return small_alloc(bytes);
    }
    auto it = pool.derived.find(tag);
    if (it != pool.derived.end() && it->second.second == bytes) {
        *fresh = false;
        return it->second.first;
    }
    if (it != pool.derived.end()) {   // same tag, other size: a different model behind the same names -- replace
        be.free(it->second.first);
        pool.bytes -= it->second.second;
    }
    void* p = be.malloc(bytes);
    pool.derived[tag] = {p, bytes};
    pool.bytes += bytes;
    *fresh = true;
    return p;
}

int Plan::ensure_dense(int v) {
    if (vals[v].ld == 0) return v;
    if (vals[v].as_dense >= 0) return vals[v].as_dense;
    const Shape s = vals[v].shape;
    int o = new_val("", s, vals[v].dtype, vals[v].lay, vals[v].batched);
    const long cols = s.back(), rows = total_elems(v) / cols, ld = vals[v].ld;
    const int es = (int)esize(vals[v].dtype);
    add_step("dense " + vals[v].name, {v}, {o}, [this, v, o, rows, cols, ld, es] {
        be.check(be.api.osg_copy_2d(be.ctx, es, ptr(v), ld, 0, ptr(o), cols, 0, rows, cols), "osg_copy_2d");
    });
    vals[v].as_dense = o;
    share_q(o, v);
    return o;
}

void Plan::add_step(const std::string& what, std::vector<int> reads, std::vector<int> writes, std::function<void()> fn) {
    Step s;
    s.what = what;
    s.run = std::move(fn);
    s.reads = std::move(reads);
    s.writes = std::move(writes);
    steps.push_back(std::move(s));
}

// NHWC -> logical (NCHW) row-major copy (the reference does this transpose on the host, :2922-2928)
int Plan::ensure_plain(int v) {
    if (vals[v].lay == Lay::plain) return v;
    if (vals[v].as_plain >= 0) return vals[v].as_plain;
    const Shape s = vals[v].shape;  // [n,C,H,W]
    int o = new_val("", s, vals[v].dtype, Lay::plain, vals[v].batched);
    long b = (vals[v].batched ? N : 1) * s[0], C = s[1], HW = s[2] * s[3];
    int es = (int)esize(vals[v].dtype);
    add_step("to_nchw " + vals[v].name, {v}, {o}, [this, v, o, b, C, HW, es] {
        long shape[3] = {b, HW, C};
        int perm[3] = {0, 2, 1};
        be.check(be.api.osg_transpose(be.ctx, es, ptr(v), ptr(o), 3, shape, perm), "osg_transpose");
    });
    vals[v].as_plain = o;
    share_q(o, v);
    return o;
}

int Plan::ensure_nhwc(int v) {
    if (vals[v].lay == Lay::nhwc) return v;
    if (vals[v].as_nhwc >= 0) return vals[v].as_nhwc;
    const Shape s = vals[v].shape;
    if (s.size() != 4) throw std::invalid_argument("Model::get_tensor_data: layout is nhwc but invalid shape.");
    long b = (vals[v].batched ? N : 1) * s[0], C = s[1], HW = s[2] * s[3];
    if (C == 1 || HW == 1) {  // identical memory image
        int o = alias(v, s, Lay::nhwc);
        vals[v].as_nhwc = o;
        return o;
    }
    int o = new_val("", s, vals[v].dtype, Lay::nhwc, vals[v].batched);
    int es = (int)esize(vals[v].dtype);
    add_step("to_nhwc " + vals[v].name, {v}, {o}, [this, v, o, b, C, HW, es] {
        long shape[3] = {b, C, HW};
        int perm[3] = {0, 2, 1};
        be.check(be.api.osg_transpose(be.ctx, es, ptr(v), ptr(o), 3, shape, perm), "osg_transpose");
    });
    vals[v].as_nhwc = o;
    share_q(o, v);
    return o;
}

// ======================================================================================================================
// The lowering proper lives in a helper class so the per-op functions can share state tersely.
// ======================================================================================================================
struct Lowering {
    Plan& P;
    Model& m;
    HipBackend& be;
    long N;
    std::unordered_map<std::string, int> uses;                    // activation name -> number of consumer ops
    std::unordered_map<std::string, int> producer;                // activation name -> op index
    std::unordered_map<std::string, std::vector<int>> consumers;  // activation name -> consumer op indices
    std::vector<char> dead;
    std::map<std::string, int> const_cache;             // file name + dtype -> val

    explicit Lowering(Plan& p) : P(p), m(p.m), be(p.be), N(p.N) {}

    std::vector<Operation>& ops() { return P.ops; }
    Val& V(int v) { return P.vals[v]; }
    osg_dtype act_dtype() const { return OSG_F16; }

    // ------------------------------------------------------------------------------------------------------------------
    // Phase B: pull every weight occurrence through the WeightsProvider in model order and make it resident.
    // dtype policy (reference get_tensor_data :2885-2909): u8 -> dequantised; f16/f32 -> the arithmetic type, except
    // operands the reference forces to float (InstanceNormalization scale/bias :4802, Pow exponent :5485, Resize scales).
    // ------------------------------------------------------------------------------------------------------------------
    static bool wants_f32(const Operation& op, size_t idx) {
        if (op.m_type == "InstanceNormalization" && (idx == 1 || idx == 2)) return true;
        if (op.m_type == "Resize") return true;
        if (op.m_type == "Pow" && idx == 1) return true;
        return false;
    }

    template <typename T>
    tensor_vector<T> fetch(WeightsProvider* wp, const std::string& fn) {
        if constexpr (std::is_same_v<T, uint8_t>) return wp->get_uint8(fn);
        else if constexpr (std::is_same_v<T, uint16_t>) return wp->get_float16(fn);
        else if constexpr (std::is_same_v<T, float>) return wp->get_float32(fn);
        else return wp->get_int64(fn);
    }

    // what one weight occurrence resolves to: the file actually read, its layout / shape, the dtype it is kept in on the device
    struct Occ { std::string fn, key; Shape shape; Lay lay; osg_dtype want; long count; };
    Occ resolve(const Operation& op, size_t i, const Tensor& t, TensorDataType ty) const {
        Occ o;
        o.fn = t.m_name;
        o.shape = to_shape(t.m_shape);
        o.lay = Lay::plain;
        auto pos = o.fn.find("_nchw.bin");
        if (pos != std::string::npos) {
            // conv weight: model.txt names the OIHW file, the runtime loads the OHWI twin (reference :2666-2692)
            if (o.shape.size() == 3) o.shape.push_back(1);  // Conv1D lifted to 2-D
            if (o.shape.size() != 4) throw std::invalid_argument("Model::get_tensor_data: layout is nhwc but invalid shape.");
            o.fn = o.fn.substr(0, pos) + "_nhwc.bin";
            o.lay = Lay::nhwc;
        }
        const bool f32 = wants_f32(op, i) || !P.fp16 || P.u8;
        // W8A16: the weight operand of a contraction stays uint8 when the on-chip dequantising kernels take its shape;
        // W8A8 (m_use_uint8_arithmetic): every uint8 weight stays uint8 -- the integer kernels consume the codes
        bool keep_u8 = false;
        if (ty == TensorDataType::uint8 && P.u8) keep_u8 = true;
        else if (ty == TensorDataType::uint8 && !f32) {
            if (P.w8_resident && i == 1) {
                if (op.m_type == "Conv") keep_u8 = o.shape.size() == 4 && o.shape[1] % 64 == 0;                      // [O,I,kh,kw]: Cin % 64
                else if (op.m_type == "MatMul" || op.m_type == "Gemm") keep_u8 = o.shape.size() == 2 && o.shape[0] % 64 == 0;   // [K,N]
            }
        }
        o.want = keep_u8 ? OSG_U8 : ty == TensorDataType::int64 ? OSG_I64 : (f32 ? OSG_F32 : OSG_F16);
        o.key = o.fn + (o.want == OSG_F32 ? "|f32" : o.want == OSG_F16 ? "|f16" : o.want == OSG_U8 ? "|u8" : "|i64");
        o.count = prod(o.shape);
        return o;
    }

    void load_weights() {
        WeightsProvider* wp = m.get_wp();
        ConstPool& pool = P.pool;
        const bool use_pool = !P.stream_weights;
        // a rebuilt plan (other batch size / input shapes / options) is served from the Model's pool: the provider is not touched
        // again -- it is exhausted by now and, with m_use_ops_cache, no longer holds the weights.  Only when a different device format
        // is wanted (fp16 arithmetic / hip_w8_resident toggled) the whole sequence is pulled once more, from a restarted provider.
        bool fetch_all = true;
        std::vector<Occ> resolved;   // of the pool check below, reused by the main loop (a plan of the LLM flow is rebuilt on every call)
        if (use_pool && pool.complete) {
            fetch_all = false;
            size_t occ = 0;
            resolved.reserve(pool.occ_types.size());
            for (auto& op : ops())
                for (size_t i = 0; i < op.m_input.size() && !fetch_all; i++) {
                    const Tensor& t = op.m_input[i];
                    if (!is_const_tensor(t)) continue;
                    if (occ >= pool.occ_types.size()) fetch_all = true;
                    else {
                        resolved.push_back(resolve(op, i, t, pool.occ_types[occ]));
                        if (!pool.base.count(resolved.back().key)) fetch_all = true;
                    }
                    occ++;
                }
            if (fetch_all) resolved.clear();
            if (fetch_all) {
                wp->on_restart();
                pool.occ_types.clear();
                pool.complete = false;
            }
        }
        size_t occ = 0;
        for (auto& op : ops())
            for (size_t i = 0; i < op.m_input.size(); i++) {
                Tensor& t = op.m_input[i];
                if (!is_const_tensor(t)) continue;
                TensorDataType ty = t.m_type;
                if (fetch_all) {
                    TensorDataType nt = wp->get_type_of_next();
                    if (nt != TensorDataType::none) ty = nt;
                    if (use_pool) pool.occ_types.push_back(ty);
                } else
                    ty = pool.occ_types[occ];
                occ++;
                const Occ o_fresh = fetch_all ? resolve(op, i, t, ty) : Occ{};
                const Occ& o = fetch_all ? o_fresh : resolved[occ - 1];
                const std::string& fn = o.fn;
                const osg_dtype want = o.want;
                const long count = o.count;
                int v = -1;
                auto it = const_cache.find(o.key);
                auto new_const = [&](void* dptr) {
                    v = P.new_val("", o.shape, want, o.lay, false);
                    Val& val = V(v);
                    val.is_const = true;
                    val.name = fn;
                    val.qscale = t.m_scale;
                    val.qzp = (int)t.m_zero_point;
                    val.dptr = dptr;
                    const_cache[o.key] = v;
                };
                if (!fetch_all) {
                    if (it != const_cache.end()) v = it->second;
                    else {
                        const ConstPool::Base& b = pool.base.at(o.key);
                        new_const(b.dptr);
                        V(v).host_f = b.host_f;
                        V(v).host_i = b.host_i;
                        V(v).host_valid = b.host_valid;
                        P.weight_bytes += b.bytes;
                    }
                    t.m_name = "#" + std::to_string(v);
                    continue;
                }
                detail::dispatch_dtype(ty, [&](auto tag) {
                    using T = typename decltype(tag)::type;
                    tensor_vector<T> data = fetch<T>(wp, fn);  // always fetched: providers serve strictly in order
                    if ((long)data.size() != count) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
                    if (it != const_cache.end()) {
                        v = it->second;
                        if (P.stream_weights) {
                            Plan::WRecipe r;
                            r.fn = fn; r.ty = ty;
                            P.recipes.push_back(r);
                        }
                        return;
                    }
                    const size_t bytes = (size_t)count * esize(want);
                    // small constants stay readable on the host for the planner (shapes, axes, eps, scales ...)
                    constexpr osg_dtype have = std::is_same_v<T, uint8_t> ? OSG_U8 : std::is_same_v<T, uint16_t> ? OSG_F16
                                               : std::is_same_v<T, float> ? OSG_F32 : OSG_I64;
                    std::vector<float> host_f;
                    std::vector<int64_t> host_i;
                    bool host_valid = false;
                    if (have == OSG_I64) {
                        host_i.assign((const int64_t*)data.data(), (const int64_t*)data.data() + count);
                        host_valid = true;
                    } else if (count <= 4096 && want != OSG_U8) {
                        host_f.resize(count);
                        for (long k = 0; k < count; k++) {
                            if constexpr (std::is_same_v<T, uint8_t>) host_f[k] = (float)((int)data[k] - (int)t.m_zero_point) * t.m_scale;
                            else if constexpr (std::is_same_v<T, uint16_t>) host_f[k] = half_to_float(data[k]);
                            else if constexpr (std::is_same_v<T, float>) host_f[k] = data[k];
                        }
                        host_valid = true;
                    }
                    auto pit = use_pool ? pool.base.find(o.key) : pool.base.end();
                    if (pit != pool.base.end()) {   // (a partial reload: this format is resident already)
                        new_const(pit->second.dptr);
                    } else if (P.budgeted && have == want && have != OSG_I64 && !host_valid && P.resident_bytes + bytes > P.vram_budget) {
                        // over the VRAM budget: this weight gets no buffer of its own -- every pass (the first one included) it is pulled from
                        // the provider again and lands in the streaming ring right before the step that reads it (Plan::restream)
                        new_const(nullptr);
                        Plan::WRecipe rec;
                        rec.val = v; rec.fn = fn; rec.ty = ty; rec.have = have; rec.want = want; rec.count = count;
                        rec.scale = t.m_scale; rec.zp = (int)t.m_zero_point;
                        rec.ring = true;
                        P.recipes.push_back(rec);
                        P.ring_weight_bytes += bytes;
                    } else {
                        new_const(be.malloc(bytes));
                        P.resident_bytes += bytes;
                        if (use_pool) {
                            ConstPool::Base b;
                            b.dptr = V(v).dptr; b.bytes = bytes; b.host_f = host_f; b.host_i = host_i; b.host_valid = host_valid;
                            pool.base[o.key] = std::move(b);
                            pool.bytes += bytes;
                        } else
                            P.owned.push_back(V(v).dptr);
                        Plan::WRecipe rec;
                        rec.val = v; rec.fn = fn; rec.ty = ty; rec.have = have; rec.want = want; rec.count = count;
                        rec.scale = t.m_scale; rec.zp = (int)t.m_zero_point;
                        if (have == want) {
                            be.check(be.api.osg_upload(be.ctx, V(v).dptr, data.data(), bytes), "osg_upload");
                        } else {
                            if (have == OSG_I64 || want == OSG_I64) throw std::invalid_argument("Model::get_tensor_data: unsupported tensor data format.");
                            void* tmp = be.malloc((size_t)count * sizeof(T));
                            be.check(be.api.osg_upload(be.ctx, tmp, data.data(), (size_t)count * sizeof(T)), "osg_upload");
                            be.check(be.api.osg_convert(be.ctx, have, want, tmp, V(v).dptr, count, t.m_scale, (int)t.m_zero_point), "osg_convert");
                            be.check(be.api.osg_sync(be.ctx), "osg_sync");
                            if (P.stream_weights) { rec.raw = tmp; P.owned.push_back(tmp); }
                            else be.free(tmp);
                        }
                        // budget mode: resident weights are fetched (providers serve strictly in order) but not re-sent.  Streamed mode: the same for the
                        // vectors small enough to stay readable on the host (biases, norm gains: <= 16 KiB each, 0.1 % of the bytes but half of the copies)
                        static const bool resend_small = getenv("OSG_STREAM_RESEND_SMALL") != nullptr;
                        rec.resident = P.budgeted || (host_valid && !resend_small);
                        if (P.stream_weights) P.recipes.push_back(rec);
                    }
                    P.weight_bytes += bytes;
                    V(v).host_f = std::move(host_f);
                    V(v).host_i = std::move(host_i);
                    V(v).host_valid = host_valid;
                });
                if (m.m_use_ops_cache && !P.stream_weights && !m.m_weights_exclusion_set.count(fn)) {
                    // resident from now on: drop the provider's host copy, like the reference's ops cache does (:4556-4569)
                    try { wp->remove(fn); } catch (const std::exception&) {}
                    m.m_weights_exclusion_set.insert(fn);
                }
                t.m_name = "#" + std::to_string(v);  // from here on the tensor names its resident val
            }
        be.check(be.api.osg_sync(be.ctx), "osg_sync");
        if (use_pool) pool.complete = true;
    }

    int const_val(const Tensor& t) const { return std::stoi(t.m_name.substr(1)); }

    // ------------------------------------------------------------------------------------------------------------------
    // graph indices for the fusion passes
    // ------------------------------------------------------------------------------------------------------------------
    void index_graph() {
        uses.clear(); producer.clear(); consumers.clear();
        uses.reserve(ops().size() * 2); producer.reserve(ops().size() * 2); consumers.reserve(ops().size() * 2);
        for (size_t i = 0; i < ops().size(); i++) {
            if (dead[i]) continue;
            for (auto& in : ops()[i].m_input)
                if (!in.m_name.empty() && in.m_type == TensorDataType::none) {
                    uses[in.m_name]++;
                    consumers[in.m_name].push_back((int)i);
                }
            for (auto& out : ops()[i].m_output) producer[out.m_name] = (int)i;
        }
        for (auto& n : P.extra_outputs) uses[n] += 1000;  // never fuse away something the caller wants to read
    }
    bool act(const Tensor& t) const { return !t.m_name.empty() && t.m_type == TensorDataType::none; }
    int prod_of(const Tensor& t) const {
        if (!act(t)) return -1;
        auto it = producer.find(t.m_name);
        return it == producer.end() ? -1 : it->second;
    }
    int use_count(const std::string& n) const { auto it = uses.find(n); return it == uses.end() ? 0 : it->second; }
    int sole_consumer(const Tensor& t) const {
        if (use_count(t.m_name) != 1) return -1;
        auto it = consumers.find(t.m_name);
        return it == consumers.end() || it->second.size() != 1 ? -1 : it->second[0];
    }
    bool is(int i, const char* type) const { return i >= 0 && !dead[i] && P.ops[i].m_type == type; }
    const Val* cval(const Tensor& t) const { return is_const_tensor(t) ? &P.vals[const_val(t)] : nullptr; }
    bool const_scalar(const Tensor& t, float* out) const {
        const Val* v = cval(t);
        if (!v || v->numel() != 1 || !v->host_valid || v->host_f.empty()) return false;
        *out = v->host_f[0];
        return true;
    }
    // the other operand of a commutative binary op
    int other(const Operation& op, const std::string& name) const { return op.m_input[0].m_name == name ? 1 : 0; }

    // a pass whose anchor op type does not occur in the (live) graph is skipped together with its re-indexing: the LLM flow re-plans on every call
    // and a 1 200-op llama graph spent 2/3 of its plan time re-indexing for passes that had nothing to match
    bool has_type(const char* type) const {
        for (size_t i = 0; i < P.ops.size(); i++)
            if (!dead[i] && P.ops[i].m_type == type) return true;
        return false;
    }
    // uint8 arithmetic: Reshape[1,G,L] -> InstanceNormalization -> Reshape[x.shape] over a 4-D tensor ==> osg.qu8.InstanceNormNHWC.  Not a fusion of
    // arithmetic (the two Reshapes carry codes and parameters through unchanged, the normalisation is one table lookup per code either way):
    // it only keeps the tensor in the convolutions' NHWC layout instead of copying it to NCHW and back (62 of the VAE decoder's launches).
    void fuse_u8_instance_norm_nhwc() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "InstanceNormalization")) continue;
            Operation& in = ops()[i];
            if (in.m_input.size() != 3 || in.m_output.size() != 1) continue;
            const int r0 = prod_of(in.m_input[0]);
            if (!is(r0, "Reshape") || use_count(in.m_input[0].m_name) != 1) continue;
            const Tensor x = ops()[r0].m_input[0];
            if (!act(x) || x.m_shape.size() != 4 || x.m_shape[0] != 1) continue;
            const auto& gs = in.m_input[0].m_shape;
            if (gs.size() != 3 || gs[0] != 1 || gs[1] == 0) continue;
            const long G = (long)gs[1], C = (long)x.m_shape[1], HW = (long)x.m_shape[2] * (long)x.m_shape[3];
            if (C % G || (long)gs[2] != (C / G) * HW || G > 56) continue;
            const int r1 = sole_consumer(in.m_output[0]);
            if (!is(r1, "Reshape") || ops()[r1].m_output[0].m_shape != x.m_shape) continue;
            Operation f;
            f.m_name = in.m_name;                 // (the op's range data is looked up under the InstanceNormalization's name)
            f.m_type = "osg.qu8.InstanceNormNHWC";
            f.m_input = {x, in.m_input[1], in.m_input[2]};
            f.m_output = {ops()[r1].m_output[0]};
            f.m_attributes = in.m_attributes;
            f.m_attributes.emplace_back("groups", std::to_string(G));
            dead[r0] = dead[i] = 1;
            ops()[r1] = std::move(f);
        }
    }

    // uint8 arithmetic: Mul(x, gamma[C]) -> Add(., beta[C]) [-> Sigmoid -> Mul(., sigmoid)] over a 4-D tensor ==> osg.qu8.AffineAct.  One pass
    // over the tensor instead of two / four; every stage re-quantises with ITS op's range exactly as the separate launches do (the fused op keeps
    // the four op names for the range lookups), so the codes are unchanged.
    void fuse_u8_affine_act() {
        auto chan_const = [&](const Tensor& t, long C) {
            const Val* v = cval(t);
            if (!v || v->dtype != OSG_U8 || v->numel() != C) return false;
            const Shape& sh = v->shape;
            return (sh.size() == 3 && sh[0] == C) || (sh.size() == 4 && sh[1] == C);
        };
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Mul")) continue;
            Operation& mu = ops()[i];
            if (mu.m_input.size() != 2 || mu.m_output.size() != 1) continue;
            int xi = -1;
            for (int k = 0; k < 2; k++)
                if (act(mu.m_input[k]) && mu.m_input[k].m_shape.size() == 4 && mu.m_input[k].m_shape[0] == 1 && chan_const(mu.m_input[1 - k], (long)mu.m_input[k].m_shape[1])) xi = k;
            if (xi < 0) continue;
            const Tensor x = mu.m_input[xi], gam = mu.m_input[1 - xi];
            const long C = (long)x.m_shape[1];
            const int ad = sole_consumer(mu.m_output[0]);
            if (!is(ad, "Add") || ops()[ad].m_input.size() != 2) continue;
            const int bidx = other(ops()[ad], mu.m_output[0].m_name);
            if (!chan_const(ops()[ad].m_input[bidx], C)) continue;
            const Tensor bet = ops()[ad].m_input[bidx];
            // optional SiLU: the Add feeds exactly a Sigmoid and the Mul that multiplies it back
            int sg = -1, m2 = -1;
            const std::string& an = ops()[ad].m_output[0].m_name;
            if (use_count(an) == 2) {
                auto it = consumers.find(an);
                if (it != consumers.end() && it->second.size() == 2) {
                    for (int c : it->second)
                        if (is(c, "Sigmoid")) sg = c;
                        else if (is(c, "Mul")) m2 = c;
                    if (sg >= 0 && m2 >= 0 && sole_consumer(ops()[sg].m_output[0]) == m2 && ops()[m2].m_input.size() == 2) {
                        const std::string &p0 = ops()[m2].m_input[0].m_name, &p1 = ops()[m2].m_input[1].m_name, &sn = ops()[sg].m_output[0].m_name;
                        if (!((p0 == an && p1 == sn) || (p0 == sn && p1 == an))) sg = m2 = -1;
                    } else
                        sg = m2 = -1;
                }
            }
            const bool silu = sg >= 0 && m2 >= 0;
            if (!silu && use_count(an) != 1) {
                // (the Add's result is read by several ops and they are not the SiLU pair: still one pass for Mul + Add)
            }
            Operation f;
            const int last = silu ? m2 : ad;
            f.m_name = ops()[last].m_name;
            f.m_type = "osg.qu8.AffineAct";
            f.m_input = {x, gam, bet};
            f.m_output = {ops()[last].m_output[0]};
            f.m_attributes = {{"mul", mu.m_name}, {"add", ops()[ad].m_name}};
            if (silu) {
                f.m_attributes.emplace_back("sigmoid", ops()[sg].m_name);
                f.m_attributes.emplace_back("mul2", ops()[m2].m_name);
            }
            // the normalisation right in front (already in its NHWC form): its table lookup rides in this pass
            const int nrm = prod_of(x);
            if (is(nrm, "osg.qu8.InstanceNormNHWC") && use_count(x.m_name) == 1) {
                f.m_type = "osg.qu8.NormAffineAct";
                f.m_input = {ops()[nrm].m_input[0], gam, bet, ops()[nrm].m_input[1], ops()[nrm].m_input[2]};
                f.m_attributes.emplace_back("norm", ops()[nrm].m_name);
                for (auto& a : ops()[nrm].m_attributes) f.m_attributes.emplace_back("norm_" + a.first, a.second);
                dead[nrm] = 1;
            }
            dead[i] = 1;
            if (silu) dead[ad] = dead[sg] = 1;
            ops()[last] = std::move(f);
        }
    }

    void run_fusions() {
        dead.assign(ops().size(), 0);
        if (P.u8 && m.m_hip_fusion_level >= 1 && has_type("InstanceNormalization")) { index_graph(); fuse_u8_instance_norm_nhwc(); }
        if (P.u8 && m.m_hip_fusion_level >= 1 && has_type("Mul") && has_type("Add")) { index_graph(); fuse_u8_affine_act(); }
        if (m.m_use_scaled_dp_attn_op && has_type("Softmax")) { index_graph(); fuse_sdpa(); }   // (a Model option of the reference, independent of hip_fusion_level)
        if (P.fusion >= 1) {
            if (m.m_requires_upcast && has_type("Pow") && has_type("ReduceMean")) { index_graph(); fuse_rms_norm(); }
            if (has_type("Neg") && has_type("Slice")) { index_graph(); fuse_rope(); }
            if (has_type("Sigmoid")) { index_graph(); fuse_silu(); }
            if (has_type("InstanceNormalization")) { index_graph(); fuse_group_norm(); }
            if (has_type("ReduceMean") && has_type("Sub")) { index_graph(); fuse_layer_norm(); }
            if (has_type("Erf")) { index_graph(); fuse_geglu(); }
        }
        if (P.fusion >= 2) {
            if (has_type("Softmax")) { index_graph(); fuse_attention(true); }
            if (has_type("MatMul")) { index_graph(); fuse_linear(); }
            if (has_type("Conv") || has_type("osg.Linear")) { index_graph(); fuse_residual(); }
            if (has_type("Conv")) { index_graph(); fuse_conv_act(); }
            if (has_type("osg.GEGLU")) { index_graph(); fuse_linear_geglu(); }
            if (has_type("osg.Attention") && has_type("osg.LayerNorm")) { index_graph(); fuse_tblock_tail(); }
            if (has_type("osg.SiLU")) { index_graph(); cse_silu(); }
            if (has_type("osg.SiLU") && has_type("Gemm")) { index_graph(); fuse_gemm_act(); }   // (after the CSE: the 22 SiLUs behind the time embedding are one by now)
            if (has_type("Conv")) { index_graph(); fuse_image_bias(); }
        } else if (m.m_fuse_ops_in_attention && has_type("Softmax")) {
            index_graph(); fuse_attention(false);
        }
        std::vector<Operation> live;
        for (size_t i = 0; i < ops().size(); i++)
            if (!dead[i]) live.push_back(std::move(ops()[i]));
        ops() = std::move(live);
    }

    // Sigmoid(x) -> Mul(x, .)   ==> osg.SiLU
    void fuse_silu() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Sigmoid")) continue;
            Operation& sg = ops()[i];
            if (sg.m_input.size() != 1 || sg.m_output.size() != 1 || !act(sg.m_input[0])) continue;
            int j = sole_consumer(sg.m_output[0]);
            if (!is(j, "Mul")) continue;
            Operation& mul = ops()[j];
            if (mul.m_input.size() != 2) continue;
            const std::string& s = sg.m_output[0].m_name;
            int o = other(mul, s);
            if (mul.m_input[o].m_name != sg.m_input[0].m_name || mul.m_input[1 - o].m_name != s) continue;
            mul.m_type = "osg.SiLU";
            Tensor x = mul.m_input[o];
            mul.m_input.clear();
            mul.m_input.push_back(x);
            dead[i] = 1;
        }
    }

    // Reshape[1,G,-1] -> InstanceNormalization(1,0) -> Reshape[n,C,H,W] -> Mul(gamma[C,1,1]) -> Add(beta[C,1,1]) [-> SiLU]
    void fuse_group_norm() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "InstanceNormalization")) continue;
            Operation& in = ops()[i];
            if (in.m_input.size() != 3 || in.m_output.size() != 1) continue;
            const Val* sc = cval(in.m_input[1]);
            const Val* bi = cval(in.m_input[2]);
            if (!sc || !bi || !sc->host_valid || !bi->host_valid) continue;
            bool unit = true;
            for (float f : sc->host_f) unit &= f == 1.0f;
            for (float f : bi->host_f) unit &= f == 0.0f;
            if (!unit) continue;
            int r0 = prod_of(in.m_input[0]);
            if (!is(r0, "Reshape") || use_count(in.m_input[0].m_name) != 1) continue;
            const Tensor& x = ops()[r0].m_input[0];
            if (!act(x) || x.m_shape.size() != 4) continue;
            if (in.m_input[0].m_shape.size() != 3 || in.m_input[0].m_shape[0] != 1) continue;
            long G = (long)in.m_input[0].m_shape[1];
            int r1 = sole_consumer(in.m_output[0]);
            if (!is(r1, "Reshape") || ops()[r1].m_output[0].m_shape != x.m_shape) continue;
            int mu = sole_consumer(ops()[r1].m_output[0]);
            if (!is(mu, "Mul")) continue;
            int gi = other(ops()[mu], ops()[r1].m_output[0].m_name);
            const Val* gam = cval(ops()[mu].m_input[gi]);
            long C = (long)x.m_shape[1];
            auto chan_shape = [&](const Val* v) {
                if (!v || v->numel() != C) return false;
                const Shape& s = v->shape;
                return (s.size() == 3 && s[0] == C) || (s.size() == 4 && s[1] == C);
            };
            if (!chan_shape(gam)) continue;
            int ad = sole_consumer(ops()[mu].m_output[0]);
            if (!is(ad, "Add")) continue;
            int bidx = other(ops()[ad], ops()[mu].m_output[0].m_name);
            if (!chan_shape(cval(ops()[ad].m_input[bidx]))) continue;
            if (C % G || x.m_shape[0] != 1) continue;
            float eps = 1e-5f;
            if (auto* e = attr(in, "epsilon")) eps = std::stof(*e);
            int last = ad;
            bool silu = false;
            int sl = sole_consumer(ops()[ad].m_output[0]);
            if (is(sl, "osg.SiLU")) { silu = true; last = sl; }
            Operation f;
            f.m_name = in.m_name + "_GroupNorm";
            f.m_type = "osg.GroupNorm";
            f.m_input = {x, ops()[mu].m_input[gi], ops()[ad].m_input[bidx]};
            f.m_output = {ops()[last].m_output[0]};
            f.m_attributes = {{"groups", std::to_string(G)}, {"epsilon", std::to_string(eps)}, {"silu", silu ? "1" : "0"}};
            char buf[64];
            snprintf(buf, sizeof buf, "%.9g", eps);
            f.m_attributes[1].second = buf;
            for (int k : {r0, (int)i, r1, mu, ad}) dead[k] = 1;
            if (silu) dead[sl] = 1;
            dead[last] = 0;
            ops()[last] = std::move(f);
        }
    }

    // ReduceMean -> Sub -> Pow(2) -> ReduceMean -> Add(eps) -> Sqrt -> Div -> Mul(gamma) -> Add(beta)
    void fuse_layer_norm() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "ReduceMean")) continue;
            Operation& rm = ops()[i];
            if (rm.m_input.size() != 1 || !act(rm.m_input[0])) continue;
            const Tensor x = rm.m_input[0];
            // both reductions must run over the LAST axis with keepdims = 1 (anything else is not a LayerNorm over the channels)
            auto last_axis_keepdims = [&](const Operation& r) {
                bool ax_ok = false, kd_ok = true;
                for (auto& a : r.m_attributes) {
                    if (a.first == "axes") {
                        auto ax = int_list(a.second);
                        ax_ok = ax.size() == 1 && !r.m_input.empty() && (ax[0] == -1 || ax[0] == (int)r.m_input[0].m_shape.size() - 1);
                    } else if (a.first == "keepdims") kd_ok = std::stoi(a.second) == 1;
                    else return false;
                }
                return ax_ok && kd_ok;
            };
            if (!last_axis_keepdims(rm)) continue;
            int sub = sole_consumer(rm.m_output[0]);
            if (!is(sub, "Sub") || ops()[sub].m_input[0].m_name != x.m_name || ops()[sub].m_input[1].m_name != rm.m_output[0].m_name) continue;
            const std::string d = ops()[sub].m_output[0].m_name;
            if (use_count(d) != 2) continue;
            auto& dc = consumers[d];
            int pw = -1, dv = -1;
            for (int c : dc) { if (is(c, "Pow")) pw = c; if (is(c, "Div")) dv = c; }
            if (pw < 0 || dv < 0 || ops()[pw].m_input[0].m_name != d || ops()[dv].m_input[0].m_name != d) continue;
            float p = 0;
            if (!const_scalar(ops()[pw].m_input[1], &p) || p != 2.0f) continue;
            int rm2 = sole_consumer(ops()[pw].m_output[0]);
            if (!is(rm2, "ReduceMean") || !last_axis_keepdims(ops()[rm2])) continue;
            int ae = sole_consumer(ops()[rm2].m_output[0]);
            if (!is(ae, "Add")) continue;
            float eps = 0;
            if (!const_scalar(ops()[ae].m_input[other(ops()[ae], ops()[rm2].m_output[0].m_name)], &eps)) continue;
            int sq = sole_consumer(ops()[ae].m_output[0]);
            if (!is(sq, "Sqrt")) continue;
            if (sole_consumer(ops()[sq].m_output[0]) != dv || ops()[dv].m_input[1].m_name != ops()[sq].m_output[0].m_name) continue;
            int mu = sole_consumer(ops()[dv].m_output[0]);
            if (!is(mu, "Mul")) continue;
            int gi = other(ops()[mu], ops()[dv].m_output[0].m_name);
            const Val* gam = cval(ops()[mu].m_input[gi]);
            long C = (long)x.m_shape.back();
            if (!gam || gam->numel() != C || gam->shape.back() != C) continue;
            int ad = sole_consumer(ops()[mu].m_output[0]);
            if (!is(ad, "Add")) continue;
            int bidx = other(ops()[ad], ops()[mu].m_output[0].m_name);
            const Val* bet = cval(ops()[ad].m_input[bidx]);
            if (!bet || bet->numel() != C || bet->shape.back() != C) continue;
            Operation f;
            f.m_name = rm.m_name + "_LayerNorm";
            f.m_type = "osg.LayerNorm";
            f.m_input = {x, ops()[mu].m_input[gi], ops()[ad].m_input[bidx]};
            f.m_output = {ops()[ad].m_output[0]};
            char buf[64];
            snprintf(buf, sizeof buf, "%.9g", eps);
            f.m_attributes = {{"epsilon", buf}};
            for (int k : {(int)i, sub, pw, rm2, ae, sq, dv, mu}) dead[k] = 1;
            ops()[ad] = std::move(f);
        }
    }

    bool slice_last_range(const Operation& sl, long* b, long* e) const {
        if (sl.m_input.size() < 3) return false;
        const Val* s = cval(sl.m_input[1]);
        const Val* en = cval(sl.m_input[2]);
        if (!s || !en || s->host_i.size() != 1 || en->host_i.size() != 1) return false;
        long rank = (long)sl.m_input[0].m_shape.size(), dim = (long)sl.m_input[0].m_shape.back();
        if (sl.m_input.size() > 3) {
            const Val* ax = cval(sl.m_input[3]);
            if (!ax || ax->host_i.size() != 1) return false;
            long a = ax->host_i[0];
            if (a < 0) a += rank;
            if (a != rank - 1) return false;
        } else if (rank != 1) return false;
        if (sl.m_input.size() > 4) {
            const Val* st = cval(sl.m_input[4]);
            if (!st || st->host_i.size() != 1 || st->host_i[0] != 1) return false;
        }
        long bb = s->host_i[0], ee = en->host_i[0];
        if (bb < 0) bb += dim;
        if (ee < 0) ee += dim;
        if (ee > dim) ee = dim;
        *b = bb; *e = ee;
        return true;
    }

    // p -> Slice(0:C)=val, Slice(C:2C)=gate ; gate -> Div(sqrt2) -> Erf -> Add(1) -> Mul(gate,.) -> Mul(.,0.5) -> Mul(val,.)
    void fuse_geglu() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Erf")) continue;
            int dv = prod_of(ops()[i].m_input[0]);
            if (!is(dv, "Div") || use_count(ops()[i].m_input[0].m_name) != 1) continue;
            float c = 0;
            if (!const_scalar(ops()[dv].m_input[1], &c) || std::fabs(c - 1.41421356f) > 2e-3f) continue;
            const Tensor gate = ops()[dv].m_input[0];
            int gs = prod_of(gate);
            if (!is(gs, "Slice") || use_count(gate.m_name) != 2) continue;
            int a1 = sole_consumer(ops()[i].m_output[0]);
            if (!is(a1, "Add")) continue;
            float one = 0;
            if (!const_scalar(ops()[a1].m_input[other(ops()[a1], ops()[i].m_output[0].m_name)], &one) || one != 1.0f) continue;
            int m1 = sole_consumer(ops()[a1].m_output[0]);
            if (!is(m1, "Mul") || ops()[m1].m_input[other(ops()[m1], ops()[a1].m_output[0].m_name)].m_name != gate.m_name) continue;
            int m2 = sole_consumer(ops()[m1].m_output[0]);
            if (!is(m2, "Mul")) continue;
            float half = 0;
            if (!const_scalar(ops()[m2].m_input[other(ops()[m2], ops()[m1].m_output[0].m_name)], &half) || half != 0.5f) continue;
            int m3 = sole_consumer(ops()[m2].m_output[0]);
            if (!is(m3, "Mul")) continue;
            const Tensor& val = ops()[m3].m_input[other(ops()[m3], ops()[m2].m_output[0].m_name)];
            int vs = prod_of(val);
            if (!is(vs, "Slice") || use_count(val.m_name) != 1) continue;
            const Tensor p = ops()[gs].m_input[0];
            if (!act(p) || ops()[vs].m_input[0].m_name != p.m_name || use_count(p.m_name) != 2) continue;
            long C2 = (long)p.m_shape.back(), b0, e0, b1, e1;
            if (C2 % 2 || !slice_last_range(ops()[vs], &b0, &e0) || !slice_last_range(ops()[gs], &b1, &e1)) continue;
            if (b0 != 0 || e0 != C2 / 2 || b1 != C2 / 2 || e1 != C2) continue;
            Operation f;
            f.m_name = ops()[i].m_name + "_GEGLU";
            f.m_type = "osg.GEGLU";
            f.m_input = {p};
            f.m_output = {ops()[m3].m_output[0]};
            for (int k : {(int)i, dv, gs, vs, a1, m1, m2}) dead[k] = 1;
            ops()[m3] = std::move(f);
        }
    }

    // follows Reshape[1,T,h,d] -> Transpose(0,2,1,3) -> Reshape[h,T,d] backwards from `t`; returns the projection tensor
    bool head_split_source(const Tensor& t, Tensor* src, long* h, long* d, std::vector<int>* chain) {
        int r1 = prod_of(t);
        if (!is(r1, "Reshape") || use_count(t.m_name) != 1 || t.m_shape.size() != 3) return false;
        const Tensor& a = ops()[r1].m_input[0];
        int tp = prod_of(a);
        if (!is(tp, "Transpose") || use_count(a.m_name) != 1) return false;
        auto* pm = attr(ops()[tp], "perm");
        if (!pm || int_list(*pm) != std::vector<int>{0, 2, 1, 3}) return false;
        const Tensor& b = ops()[tp].m_input[0];
        int r0 = prod_of(b);
        if (!is(r0, "Reshape") || use_count(b.m_name) != 1 || b.m_shape.size() != 4 || b.m_shape[0] != 1) return false;
        const Tensor& s = ops()[r0].m_input[0];
        if (!act(s) || s.m_shape.size() != 3 || s.m_shape[0] != 1) return false;
        long T = (long)b.m_shape[1];
        *h = (long)b.m_shape[2];
        *d = (long)b.m_shape[3];
        if ((long)s.m_shape[1] != T || (long)s.m_shape[2] != *h * *d) return false;
        if ((long)t.m_shape[0] != *h || (long)t.m_shape[1] != T || (long)t.m_shape[2] != *d) return false;
        *src = s;
        chain->insert(chain->end(), {r1, tp, r0});
        return true;
    }

    // Pow(x, 2) -> ReduceMean(-1, keepdims) -> Add(eps) -> Sqrt -> Div(1, .) -> Mul(x, .) -> Mul(w, .)  ==> osg.RMSNorm, when m_requires_upcast flags
    // all seven ops (the LLM app's layer norms, src/llm.cpp:379-383): every intermediate then stays fp32 in the reference (each is the sole operand
    // of the next op), so the chain is fp32 arithmetic from the f16 input to one rounding -- which is what the fused kernel does.  Unflagged
    // chains keep their op-by-op f16 roundings.
    bool upcast_flag(const Operation& op) const { return P.fp16 && !P.u8 && m.m_requires_upcast && m.m_requires_upcast(op.m_type, op.m_name); }
    void fuse_rms_norm() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Pow")) continue;
            Operation& pw = ops()[i];
            float two = 0.f, eps = 0.f, one = 0.f;
            if (pw.m_input.size() != 2 || !act(pw.m_input[0]) || !const_scalar(pw.m_input[1], &two) || two != 2.0f) continue;
            const Tensor x = pw.m_input[0];
            const int rm = sole_consumer(pw.m_output[0]);
            if (!is(rm, "ReduceMean")) continue;
            {
                auto* ax = attr(ops()[rm], "axes");
                auto* kd = attr(ops()[rm], "keepdims");
                if (!ax || int_list(*ax) != std::vector<int>{-1} || !kd || *kd != "1") continue;
            }
            const int ad = sole_consumer(ops()[rm].m_output[0]);
            if (!is(ad, "Add") || ops()[ad].m_input.size() != 2) continue;
            if (!const_scalar(ops()[ad].m_input[other(ops()[ad], ops()[rm].m_output[0].m_name)], &eps)) continue;
            const int sq = sole_consumer(ops()[ad].m_output[0]);
            if (!is(sq, "Sqrt")) continue;
            const int dv = sole_consumer(ops()[sq].m_output[0]);
            if (!is(dv, "Div") || ops()[dv].m_input.size() != 2 || ops()[dv].m_input[1].m_name != ops()[sq].m_output[0].m_name ||
                !const_scalar(ops()[dv].m_input[0], &one) || one != 1.0f)
                continue;
            const int m0 = sole_consumer(ops()[dv].m_output[0]);
            if (!is(m0, "Mul") || ops()[m0].m_input.size() != 2) continue;
            if (ops()[m0].m_input[other(ops()[m0], ops()[dv].m_output[0].m_name)].m_name != x.m_name) continue;
            const int m1 = sole_consumer(ops()[m0].m_output[0]);
            if (!is(m1, "Mul") || ops()[m1].m_input.size() != 2) continue;
            const Tensor w = ops()[m1].m_input[other(ops()[m1], ops()[m0].m_output[0].m_name)];
            const Val* wv = cval(w);
            if (!wv || wv->dtype != OSG_F16 || wv->shape.size() != 1) continue;
            bool all_up = true;
            for (int k : {(int)i, rm, ad, sq, dv, m0, m1}) all_up &= upcast_flag(ops()[k]);
            if (!all_up) continue;
            Operation f;
            f.m_name = ops()[m1].m_name + "_RMSNorm";
            f.m_type = "osg.RMSNorm";
            f.m_input = {x, w};
            f.m_output = {ops()[m1].m_output[0]};
            char buf[64];
            snprintf(buf, sizeof buf, "%.9g", eps);
            f.m_attributes = {{"epsilon", buf}};
            for (int k : {(int)i, rm, ad, sq, dv, m0}) dead[k] = 1;
            ops()[m1] = std::move(f);
        }
    }

    // x * cos + Concat(Neg(Slice(x, d/2:d)), Slice(x, 0:d/2)) * sin  ==> osg.RoPE (HF rotate_half; same f16 roundings as the seven ops: bit-identical)
    void fuse_rope() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Add")) continue;
            Operation& add = ops()[i];
            if (add.m_input.size() != 2) continue;
            const int ma = prod_of(add.m_input[0]), mb = prod_of(add.m_input[1]);
            if (!is(ma, "Mul") || !is(mb, "Mul") || use_count(add.m_input[0].m_name) != 1 || use_count(add.m_input[1].m_name) != 1) continue;
            bool done = false;
            for (int swap = 0; swap < 2 && !done; swap++) {
                const Operation& mx = ops()[swap ? mb : ma];      // x * cos
                const Operation& mr = ops()[swap ? ma : mb];      // rotate_half(x) * sin
                if (mx.m_input.size() != 2 || mr.m_input.size() != 2) continue;
                for (int rs = 0; rs < 2 && !done; rs++) {
                    const Tensor rot = mr.m_input[rs], sin_t = mr.m_input[1 - rs];
                    const int cc = prod_of(rot);
                    if (!is(cc, "Concat") || use_count(rot.m_name) != 1 || ops()[cc].m_input.size() != 2) continue;
                    auto* cax = attr(ops()[cc], "axis");
                    if (!cax) continue;
                    const int ng = prod_of(ops()[cc].m_input[0]), s1 = prod_of(ops()[cc].m_input[1]);
                    if (!is(ng, "Neg") || !is(s1, "Slice") || use_count(ops()[cc].m_input[0].m_name) != 1 || use_count(ops()[cc].m_input[1].m_name) != 1) continue;
                    const int s2 = prod_of(ops()[ng].m_input[0]);
                    if (!is(s2, "Slice") || use_count(ops()[ng].m_input[0].m_name) != 1) continue;
                    const Tensor x = ops()[s1].m_input[0];
                    if (!act(x) || ops()[s2].m_input[0].m_name != x.m_name || x.m_shape.empty()) continue;
                    const long d = (long)x.m_shape.back(), rank = (long)x.m_shape.size();
                    if (d <= 0 || d % 2) continue;
                    const int caxis = std::stoi(*cax);
                    if (caxis != -1 && caxis != rank - 1) continue;
                    auto slice_is = [&](int si, long b0, long e0) {
                        const Operation& so = ops()[si];
                        if (so.m_input.size() < 4) return false;
                        const Val *st = cval(so.m_input[1]), *en = cval(so.m_input[2]), *ax = cval(so.m_input[3]);
                        if (!st || !en || !ax || st->host_i.size() != 1 || en->host_i.size() != 1 || ax->host_i.size() != 1) return false;
                        if (so.m_input.size() > 4 && !so.m_input[4].m_name.empty()) {
                            const Val* sp = cval(so.m_input[4]);
                            if (!sp || sp->host_i.size() != 1 || sp->host_i[0] != 1) return false;
                        }
                        const long a = ax->host_i[0];
                        return (a == -1 || a == rank - 1) && st->host_i[0] == b0 && (en->host_i[0] == e0 || (e0 == d && en->host_i[0] >= d));
                    };
                    if (!slice_is(s1, 0, d / 2) || !slice_is(s2, d / 2, d)) continue;
                    const int xi = mx.m_input[0].m_name == x.m_name ? 0 : mx.m_input[1].m_name == x.m_name ? 1 : -1;
                    if (xi < 0) continue;
                    const Tensor cos_t = mx.m_input[1 - xi];
                    Operation f;
                    f.m_name = add.m_name + "_RoPE";
                    f.m_type = "osg.RoPE";
                    f.m_input = {x, cos_t, sin_t};
                    f.m_output = {add.m_output[0]};
                    for (int k : {ma, mb, cc, ng, s1, s2}) dead[k] = 1;
                    ops()[i] = std::move(f);
                    done = true;
                }
            }
        }
    }

    // m_use_scaled_dp_attn_op: the reference's ScaledDotProductAttention rewrite (src/onnxstream.cpp:3635-3755), both forms:
    //   Transpose(k) -> MatMul(q, .) -> Div(., s) -> Add(., mask) -> Softmax(-1) -> MatMul(., v)              scale = f16(1 / s)
    //   Transpose(k) -> Mul(., s2); Mul(q, s) -> MatMul -> Add(., mask) -> Softmax(-1) -> MatMul(., v)         scale = f16(s2 * s)
    // every intermediate with exactly one consumer (the reference's m_intermediate_refs == 1 checks).  The reference matches the ops
    // as CONSECUTIVE queue entries; here they are matched through the graph (the same chains, whatever else the exporter interleaved).
    void fuse_sdpa() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Softmax")) continue;
            Operation& sm = ops()[i];
            auto* ax = attr(sm, "axis");
            if (!ax || *ax != "-1" || sm.m_attributes.size() != 1 || sm.m_input.size() != 1 || sm.m_output.size() != 1) continue;
            const int add = prod_of(sm.m_input[0]);
            if (!is(add, "Add") || use_count(sm.m_input[0].m_name) != 1 || ops()[add].m_input.size() != 2) continue;
            const Tensor mask = ops()[add].m_input[1];
            const Tensor scores = ops()[add].m_input[0];
            const int pre = prod_of(scores);
            if (pre < 0 || use_count(scores.m_name) != 1) continue;
            int mm0 = -1, tp = -1;
            std::vector<int> chain = {(int)i, add};
            Tensor q, s_t, s2_t;
            if (is(pre, "Div") && ops()[pre].m_input.size() == 2) {
                s_t = ops()[pre].m_input[1];
                mm0 = prod_of(ops()[pre].m_input[0]);
                if (!is(mm0, "MatMul") || use_count(ops()[pre].m_input[0].m_name) != 1 || ops()[mm0].m_input.size() != 2) continue;
                q = ops()[mm0].m_input[0];
                tp = prod_of(ops()[mm0].m_input[1]);
                if (tp < 0 || use_count(ops()[mm0].m_input[1].m_name) != 1) continue;
                chain.insert(chain.end(), {pre, mm0});
            } else if (is(pre, "MatMul") && ops()[pre].m_input.size() == 2) {
                mm0 = pre;
                const int mul0 = prod_of(ops()[mm0].m_input[0]), mul1 = prod_of(ops()[mm0].m_input[1]);
                if (!is(mul0, "Mul") || !is(mul1, "Mul") || ops()[mul0].m_input.size() != 2 || ops()[mul1].m_input.size() != 2) continue;
                if (use_count(ops()[mm0].m_input[0].m_name) != 1 || use_count(ops()[mm0].m_input[1].m_name) != 1) continue;
                q = ops()[mul0].m_input[0];
                s_t = ops()[mul0].m_input[1];
                s2_t = ops()[mul1].m_input[1];
                tp = prod_of(ops()[mul1].m_input[0]);
                if (tp < 0 || use_count(ops()[mul1].m_input[0].m_name) != 1) continue;
                chain.insert(chain.end(), {mm0, mul0, mul1});
            } else continue;
            auto* pm = is(tp, "Transpose") ? attr(ops()[tp], "perm") : nullptr;
            if (!pm || int_list(*pm) != std::vector<int>{0, 1, 3, 2} || ops()[tp].m_input.size() != 1) continue;
            const Tensor k = ops()[tp].m_input[0];
            const int mm1 = sole_consumer(sm.m_output[0]);
            if (!is(mm1, "MatMul") || ops()[mm1].m_input.size() != 2 || ops()[mm1].m_input[0].m_name != sm.m_output[0].m_name) continue;
            const Tensor v = ops()[mm1].m_input[1];
            if (!act(q) || !act(k) || !act(v)) continue;
            float s = 0.f, s2 = 0.f;
            auto scalar = [&](const Tensor& t, float* out) {   // "invalid shape of scale" (:7785): a scalar or a 1-element vector
                const Val* sv = cval(t);
                return sv && sv->numel() == 1 && sv->shape.size() <= 1 && const_scalar(t, out);
            };
            if (!scalar(s_t, &s) || (!s2_t.m_name.empty() && !scalar(s2_t, &s2))) continue;
            // fp16 arithmetic: the constants are f16 by the time the op sees them, the product / reciprocal is formed in fp32 and rounded to f16 (:7840-7862)
            const float s16 = half_to_float(float_to_half(s));
            const float val = s2_t.m_name.empty() ? 1.0f / s16 : half_to_float(float_to_half(s2)) * s16;
            const float scale = half_to_float(float_to_half(val));
            Operation f;
            f.m_name = ops()[tp].m_name + "_ScaledDotProductAttention";
            f.m_type = "ScaledDotProductAttention";
            f.m_input = {q, k, mask, v};
            f.m_output = {ops()[mm1].m_output[0]};
            char buf[64];
            snprintf(buf, sizeof buf, "%.9g", scale);
            f.m_attributes = {{"scale", buf}};
            chain.push_back(tp);
            for (int c : chain) dead[c] = 1;
            ops()[mm1] = std::move(f);
        }
    }

    // MatMul(q,kT) -> [Mul(scale)] -> Softmax(-1) -> MatMul(.,v)
    //   full=false: the reference's own AttentionFusedOps rewrite (:3576-3633) -- operands stay [h,T,d]/[h,d,Tk]
    //   full=true : additionally folds the head split / merge Reshape+Transpose chains: reads Q,K,V straight from the
    //               [1,T,h*d] projection outputs and writes [1,T,h*d]
    void fuse_attention(bool full) {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Softmax")) continue;
            Operation& sm = ops()[i];
            auto* ax = attr(sm, "axis");
            if (!ax || *ax != "-1" || sm.m_attributes.size() != 1 || sm.m_input.size() != 1) continue;
            int pre = prod_of(sm.m_input[0]);
            if (pre < 0 || use_count(sm.m_input[0].m_name) != 1) continue;
            int mul = -1, mm0 = pre;
            Tensor scale_t;
            if (is(pre, "Mul")) {
                mul = pre;
                if (ops()[mul].m_input.size() != 2) continue;
                mm0 = prod_of(ops()[mul].m_input[0]);
                if (mm0 < 0 || use_count(ops()[mul].m_input[0].m_name) != 1) continue;
                scale_t = ops()[mul].m_input[1];
            }
            if (!is(mm0, "MatMul")) continue;
            int mm1 = sole_consumer(sm.m_output[0]);
            if (!is(mm1, "MatMul") || ops()[mm1].m_input[0].m_name != sm.m_output[0].m_name) continue;
            const Tensor q = ops()[mm0].m_input[0], kt = ops()[mm0].m_input[1], v = ops()[mm1].m_input[1];
            if (!act(q) || !act(kt) || !act(v)) continue;
            float scale = 1.0f;
            if (mul >= 0) {
                const Val* sv = cval(scale_t);
                if (!sv || !sv->shape.empty() || !const_scalar(scale_t, &scale)) continue;  // "s must be a scalar" (:6723)
            }
            std::vector<int> chain = {mm0, (int)i};
            if (mul >= 0) chain.push_back(mul);
            if (full) {
                Tensor qs, ks, vs;
                long h, d, h2, d2, h3, d3;
                int ktp = prod_of(kt);
                auto* pm = ktp >= 0 ? attr(ops()[ktp], "perm") : nullptr;
                if (!is(ktp, "Transpose") || !pm || int_list(*pm) != std::vector<int>{0, 2, 1} || use_count(kt.m_name) != 1) goto plain;
                {
                    std::vector<int> c2 = chain;
                    c2.push_back(ktp);
                    if (!head_split_source(q, &qs, &h, &d, &c2) || !head_split_source(ops()[ktp].m_input[0], &ks, &h2, &d2, &c2) ||
                        !head_split_source(v, &vs, &h3, &d3, &c2) || h != h2 || h != h3 || d != d2 || d != d3)
                        goto plain;
                    // merge: Reshape[1,h,T,d] -> Transpose(0,2,1,3) -> Reshape[1,T,h*d]
                    int o0 = sole_consumer(ops()[mm1].m_output[0]);
                    if (!is(o0, "Reshape")) goto plain;
                    int o1 = sole_consumer(ops()[o0].m_output[0]);
                    auto* pm2 = o1 >= 0 ? attr(ops()[o1], "perm") : nullptr;
                    if (!is(o1, "Transpose") || !pm2 || int_list(*pm2) != std::vector<int>{0, 2, 1, 3}) goto plain;
                    int o2 = sole_consumer(ops()[o1].m_output[0]);
                    if (!is(o2, "Reshape")) goto plain;
                    const auto& os = ops()[o2].m_output[0].m_shape;
                    if (os.size() != 3 || os[0] != 1 || (long)os[1] != (long)q.m_shape[1] || (long)os[2] != h * d) goto plain;
                    Operation f;
                    f.m_name = ops()[mm0].m_name + "_Attention";
                    f.m_type = "osg.Attention";
                    f.m_input = {qs, ks, vs};
                    f.m_output = {ops()[o2].m_output[0]};
                    char buf[64];
                    snprintf(buf, sizeof buf, "%.9g", scale);
                    f.m_attributes = {{"heads", std::to_string(h)}, {"scale", buf}};
                    for (int k : c2) dead[k] = 1;
                    for (int k : {mm1, o0, o1}) dead[k] = 1;
                    ops()[o2] = std::move(f);
                    continue;
                }
            }
        plain:
            {
                Operation f;
                f.m_name = ops()[mm0].m_name + "_AttentionFusedOps";
                f.m_type = "AttentionFusedOps";
                f.m_input = {q, kt, mul >= 0 ? scale_t : Tensor(), v};
                f.m_output = {ops()[mm1].m_output[0]};
                for (int k : chain) dead[k] = 1;
                ops()[mm1] = std::move(f);
            }
        }
    }

    // osg.Linear(x, W[K,2C], b) -> osg.GEGLU  ==> the GEGLU rides in the GEMM epilogue (value/gate columns pair-interleaved at plan time)
    void fuse_linear_geglu() {
        if (P.stream_weights) return;   // needs a re-ordered resident copy of the weight
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "osg.Linear")) continue;
            Operation& op = ops()[i];
            if (attr(op, "osg_residual") || op.m_output.size() != 1) continue;
            const Val* w = cval(op.m_input[1]);
            if (!w || w->shape.size() != 2 || w->dtype != OSG_F16 || w->shape[1] % 32 || w->shape[0] % 64) continue;
            if (op.m_input.size() > 2 && !op.m_input[2].m_name.empty()) {
                const Val* b = cval(op.m_input[2]);
                if (!b || b->dtype != OSG_F16) continue;
            }
            int ge = sole_consumer(op.m_output[0]);
            if (!is(ge, "osg.GEGLU")) continue;
            Operation f = op;
            f.m_attributes.emplace_back("osg_geglu", "1");
            f.m_output = {ops()[ge].m_output[0]};
            dead[i] = 1;
            ops()[ge] = std::move(f);
        }
    }

    // The row-local tail of a BasicTransformerBlock ==> osg.TBlockTail (one launch, osg_tblock_tail / osg_tchain.hip):
    //   osg.Linear(a1, Wo1, bo1, +x0) = x1 -> osg.LayerNorm -> osg.Linear(Wq2) = q -> osg.Attention(q, k, v) -> osg.Linear(Wo2, bo2, +x1) = x2
    //   -> osg.LayerNorm -> osg.Linear(W1, b1, geglu) -> osg.Linear(W2, b2, +x2) = x3 [-> Reshape -> Transpose(0,3,1,2) -> Conv 1x1 (+ residual) = y]
    // anchored at the cross-attention (k / v do not depend on the rows: they are projections of the text context).  Everything between a1 and x3 / y
    // must have exactly the readers the chain itself accounts for.
    void fuse_tblock_tail() {
        if (P.stream_weights || !m.m_hip_fuse_tblock || P.w8_resident) return;
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "osg.Attention")) continue;
            const Operation& at = ops()[i];
            if (at.m_input.size() != 3 || at.m_output.size() != 1) continue;
            const Tensor q = at.m_input[0], kt = at.m_input[1], vt = at.m_input[2];
            if (q.m_name == kt.m_name || q.m_name == vt.m_name) continue;                    // self-attention: q, k, v are views of one projection
            // q = Linear(LayerNorm(x1)), no residual
            const int lq = prod_of(q);
            if (!is(lq, "osg.Linear") || use_count(q.m_name) != 1 || attr(ops()[lq], "osg_residual") || attr(ops()[lq], "osg_geglu")) continue;
            const int ln2 = prod_of(ops()[lq].m_input[0]);
            if (!is(ln2, "osg.LayerNorm") || use_count(ops()[lq].m_input[0].m_name) != 1) continue;
            const Tensor x1 = ops()[ln2].m_input[0];
            const int lo1 = prod_of(x1);
            if (!is(lo1, "osg.Linear") || !attr(ops()[lo1], "osg_residual") || attr(ops()[lo1], "osg_geglu") || ops()[lo1].m_input.size() != 4 || use_count(x1.m_name) != 2) continue;
            // a2 -> Linear(+x1) = x2
            const int lo2 = sole_consumer(at.m_output[0]);
            if (!is(lo2, "osg.Linear") || !attr(ops()[lo2], "osg_residual") || attr(ops()[lo2], "osg_geglu") || ops()[lo2].m_input.size() != 4) continue;
            if (ops()[lo2].m_input[0].m_name != at.m_output[0].m_name || ops()[lo2].m_input[3].m_name != x1.m_name) continue;
            const Tensor x2 = ops()[lo2].m_output[0];
            if (use_count(x2.m_name) != 2) continue;
            int ln3 = -1, l2 = -1;
            for (int c : consumers[x2.m_name]) { if (is(c, "osg.LayerNorm")) ln3 = c; else if (is(c, "osg.Linear")) l2 = c; }
            if (ln3 < 0 || l2 < 0 || ops()[ln3].m_input[0].m_name != x2.m_name) continue;
            const int l1 = sole_consumer(ops()[ln3].m_output[0]);
            if (!is(l1, "osg.Linear") || !attr(ops()[l1], "osg_geglu") || ops()[l1].m_input[0].m_name != ops()[ln3].m_output[0].m_name) continue;
            if (sole_consumer(ops()[l1].m_output[0]) != l2 || !attr(ops()[l2], "osg_residual") || ops()[l2].m_input.size() != 4) continue;
            if (ops()[l2].m_input[0].m_name != ops()[l1].m_output[0].m_name || ops()[l2].m_input[3].m_name != x2.m_name) continue;
            // operands: f16 resident 2-D weights, f16 vectors
            auto wok = [&](const Operation& lin, long K, long Nn) {
                const Val* w = cval(lin.m_input[1]);
                return w && w->dtype == OSG_F16 && w->shape.size() == 2 && w->shape[0] == K && w->shape[1] == Nn;
            };
            auto vok = [&](const Tensor& t, long n, bool may_be_empty) {
                if (t.m_name.empty()) return may_be_empty;
                const Val* b = cval(t);
                return b && b->dtype == OSG_F16 && b->numel() == n;
            };
            const auto& xs = x1.m_shape;
            if (xs.size() != 3 || xs[0] != 1) continue;
            const long T = (long)xs[1], C = (long)xs[2], F = 4 * C;
            const Operation &O1 = ops()[lo1], &LQ = ops()[lq], &O2 = ops()[lo2], &L1 = ops()[l1], &L2 = ops()[l2];
            if (!wok(O1, C, C) || !wok(LQ, C, C) || !wok(O2, C, C) || !wok(L1, C, 2 * F) || !wok(L2, F, C)) continue;
            if (!vok(O1.m_input[2], C, true) || !vok(O2.m_input[2], C, true) || !vok(L1.m_input.size() > 2 ? L1.m_input[2] : Tensor(), 2 * F, true) || !vok(L2.m_input[2], C, true)) continue;
            if (LQ.m_input.size() > 2 && !vok(LQ.m_input[2], C, true)) continue;
            if (!vok(ops()[ln2].m_input[1], C, false) || !vok(ops()[ln2].m_input[2], C, false) || !vok(ops()[ln3].m_input[1], C, false) || !vok(ops()[ln3].m_input[2], C, false)) continue;
            if (!act(O1.m_input[0]) || !act(O1.m_input[3]) || O1.m_input[0].m_shape != xs || O1.m_input[3].m_shape != xs) continue;
            if (kt.m_shape.size() != 3 || kt.m_shape[0] != 1 || (long)kt.m_shape[2] != C || vt.m_shape != kt.m_shape) continue;
            const long heads = std::stol(*attr(at, "heads")), Tk = (long)kt.m_shape[1];
            if (!be.api.osg_tblock_tail_supported((int)(T * N), (int)T, (int)C, (int)heads, (int)Tk)) continue;
            // optional: x3 -> Reshape -> Transpose(0,3,1,2) -> Conv 1x1 (bias, + residual)
            int cv = -1, rsh = -1, trp = -1;
            {
                const int r0 = sole_consumer(L2.m_output[0]);
                const int t0 = is(r0, "Reshape") ? sole_consumer(ops()[r0].m_output[0]) : -1;
                auto* pm = t0 >= 0 ? attr(ops()[t0], "perm") : nullptr;
                const int c0 = is(t0, "Transpose") && pm && int_list(*pm) == std::vector<int>{0, 3, 1, 2} ? sole_consumer(ops()[t0].m_output[0]) : -1;
                if (is(c0, "Conv")) {
                    const Operation& co = ops()[c0];
                    const Val* w = cval(co.m_input[1]);
                    const auto& rs = ops()[r0].m_output[0].m_shape;
                    bool ok = co.m_input.size() == 4 && attr(co, "osg_residual") && !attr(co, "osg_image_bias") && !attr(co, "osg_act") &&
                              co.m_input[0].m_name == ops()[t0].m_output[0].m_name && w && w->dtype == OSG_F16 && w->shape == Shape{C, C, 1, 1} && vok(co.m_input[2], C, true) &&
                              rs.size() == 4 && rs[0] == 1 && (long)(rs[1] * rs[2]) == T && (long)rs[3] == C && act(co.m_input[3]) && co.m_input[3].m_name != co.m_input[0].m_name;
                    for (auto& a : co.m_attributes) {
                        if (a.first == "pads") { for (int v2 : int_list(a.second)) ok = ok && v2 == 0; }
                        else if (a.first == "strides" || a.first == "dilations" || a.first == "kernel_shape") { for (int v2 : int_list(a.second)) ok = ok && v2 == 1; }
                        else if (a.first == "group") ok = ok && std::stoi(a.second) == 1;
                    }
                    if (ok) { cv = c0; rsh = r0; trp = t0; }
                }
            }
            Operation f;
            f.m_name = at.m_name + "_TBlockTail";
            f.m_type = "osg.TBlockTail";
            const Tensor none;
            f.m_input = {O1.m_input[0], O1.m_input[3], O1.m_input[1], O1.m_input[2], ops()[ln2].m_input[1], ops()[ln2].m_input[2], LQ.m_input[1],
                         LQ.m_input.size() > 2 ? LQ.m_input[2] : none, kt, vt, O2.m_input[1], O2.m_input[2], ops()[ln3].m_input[1], ops()[ln3].m_input[2],
                         L1.m_input[1], L1.m_input.size() > 2 ? L1.m_input[2] : none, L2.m_input[1], L2.m_input[2]};
            f.m_attributes = {{"heads", *attr(at, "heads")}, {"scale", *attr(at, "scale")}, {"eps2", *attr(ops()[ln2], "epsilon")}, {"eps3", *attr(ops()[ln3], "epsilon")},
                              {"proj", cv >= 0 ? "1" : "0"}};
            const int last = cv >= 0 ? cv : l2;
            if (cv >= 0) {
                f.m_input.push_back(ops()[cv].m_input[1]);
                f.m_input.push_back(ops()[cv].m_input[2]);
                f.m_input.push_back(ops()[cv].m_input[3]);
            }
            f.m_output = {ops()[last].m_output[0]};
            for (int k2 : {lo1, ln2, lq, (int)i, lo2, ln3, l1, l2, rsh, trp})
                if (k2 >= 0 && k2 != last) dead[k2] = 1;
            ops()[last] = std::move(f);
        }
    }

    // identical osg.SiLU(x) ops (the 22 resnet blocks each re-activate the SAME time embedding) ==> one
    void cse_silu() {
        std::map<std::string, std::string> first;   // input name -> surviving output name
        std::map<std::string, std::string> rename;
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "osg.SiLU")) continue;
            Operation& op = ops()[i];
            if (op.m_input.size() != 1 || op.m_output.size() != 1 || !act(op.m_input[0])) continue;
            if (use_count(op.m_output[0].m_name) >= 1000) continue;  // an extra output the caller reads
            auto it = first.find(op.m_input[0].m_name);
            if (it == first.end()) first[op.m_input[0].m_name] = op.m_output[0].m_name;
            else {
                rename[op.m_output[0].m_name] = it->second;
                dead[i] = 1;
            }
        }
        if (rename.empty()) return;
        for (size_t i = 0; i < ops().size(); i++) {
            if (dead[i]) continue;
            for (auto& in : ops()[i].m_input) {
                auto it = rename.find(in.m_name);
                if (it != rename.end() && in.m_type == TensorDataType::none) in.m_name = it->second;
            }
        }
    }

    // Conv(x) -> Add(., Unsqueeze(Unsqueeze(g[1,C])))  ==> the per-image channel bias g rides in the conv epilogue
    void fuse_image_bias() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Conv")) continue;
            Operation& op = ops()[i];
            if (op.m_output.size() != 1 || attr(op, "osg_residual")) continue;
            int ad = sole_consumer(op.m_output[0]);
            if (!is(ad, "Add")) continue;
            int ti = other(ops()[ad], op.m_output[0].m_name);
            const Tensor& t = ops()[ad].m_input[ti];
            const auto& os = op.m_output[0].m_shape;
            if (!act(t) || os.size() != 4 || t.m_shape != std::vector<size_t>{1, os[1], 1, 1}) continue;
            int u1 = prod_of(t);
            if (!is(u1, "Unsqueeze") || use_count(t.m_name) != 1) continue;
            int u0 = prod_of(ops()[u1].m_input[0]);
            if (!is(u0, "Unsqueeze") || use_count(ops()[u1].m_input[0].m_name) != 1) continue;
            const Tensor g = ops()[u0].m_input[0];
            if (!act(g) || g.m_shape != std::vector<size_t>{1, os[1]} || use_count(g.m_name) != 1) continue;
            Operation f = op;
            while (f.m_input.size() < 4) f.m_input.push_back(Tensor());
            f.m_input.push_back(g);
            f.m_attributes.emplace_back("osg_image_bias", "1");
            f.m_output = {ops()[ad].m_output[0]};
            dead[i] = dead[u0] = dead[u1] = 1;
            ops()[ad] = std::move(f);
        }
    }

    // MatMul(x, W) -> Add(., b[N])  ==> osg.Linear(x, W, b)
    void fuse_linear() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "MatMul")) continue;
            Operation& mm = ops()[i];
            const Val* w = cval(mm.m_input[1]);
            if (!act(mm.m_input[0]) || !w || w->shape.size() != 2) continue;
            mm.m_type = "osg.Linear";
            int ad = sole_consumer(mm.m_output[0]);
            if (!is(ad, "Add")) continue;
            int bi = other(ops()[ad], mm.m_output[0].m_name);
            const Val* b = cval(ops()[ad].m_input[bi]);
            if (!b || b->shape.size() != 1 || b->shape[0] != w->shape[1]) continue;
            Operation f = mm;
            f.m_input.push_back(ops()[ad].m_input[bi]);
            f.m_output = {ops()[ad].m_output[0]};
            dead[i] = 1;
            ops()[ad] = std::move(f);
        }
    }

    // {Conv | osg.Linear}(..) -> Add(., r)  with r an activation of the same shape  ==> residual fused in the epilogue
    void fuse_residual() {
        for (size_t i = 0; i < ops().size(); i++) {
            const bool conv = is((int)i, "Conv"), lin = is((int)i, "osg.Linear");
            if (!conv && !lin) continue;
            Operation& op = ops()[i];
            if (op.m_output.size() != 1) continue;
            int ad = sole_consumer(op.m_output[0]);
            if (!is(ad, "Add")) continue;
            int ri = other(ops()[ad], op.m_output[0].m_name);
            const Tensor& r = ops()[ad].m_input[ri];
            if (!act(r) || r.m_shape != op.m_output[0].m_shape || r.m_name == op.m_output[0].m_name) continue;
            if (conv && op.m_output[0].m_shape.size() != 4) continue;   // (a Conv1D is lifted to 2-D inside lower_conv: its residual stays an Add)
            // the residual must already exist when the fused op runs: its producer has to precede `ad` (always true in a
            // topologically sorted file) -- the fused op takes the place of the Add.
            Operation f = op;
            if (conv && f.m_input.size() == 2) f.m_input.push_back(Tensor());  // no bias
            if (lin && f.m_input.size() == 2) f.m_input.push_back(Tensor());
            f.m_input.push_back(r);
            f.m_attributes.emplace_back("osg_residual", "1");
            f.m_output = {ops()[ad].m_output[0]};
            dead[i] = 1;
            ops()[ad] = std::move(f);
        }
    }

    // Conv(..) -> osg.SiLU  ==> the activation rides in the convolution's epilogue (the Conv -> Sigmoid -> Mul triple of every block of
    // the exported YOLOv8 graphs; in the SD UNet SiLU follows the GroupNorms instead)
    void fuse_conv_act() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Conv")) continue;
            Operation& op = ops()[i];
            if (op.m_output.size() != 1 || attr(op, "osg_act")) continue;
            int sl = sole_consumer(op.m_output[0]);
            if (!is(sl, "osg.SiLU") || ops()[sl].m_input.size() != 1) continue;
            Operation f = op;
            f.m_attributes.emplace_back("osg_act", "silu");
            f.m_output = {ops()[sl].m_output[0]};
            dead[i] = 1;
            ops()[sl] = std::move(f);
        }
    }

    // Gemm(..) -> osg.SiLU  ==> the activation rides in the GEMM's epilogue (time_embedding.linear_1 -> act, linear_2 -> the resnets' nonlinearity:
    // two launches of the UNet's time-embedding chain)
    void fuse_gemm_act() {
        for (size_t i = 0; i < ops().size(); i++) {
            if (!is((int)i, "Gemm")) continue;
            Operation& op = ops()[i];
            if (op.m_output.size() != 1 || op.m_input.size() != 3 || attr(op, "osg_act")) continue;
            const Val* w = cval(op.m_input[1]);
            if (!w || w->dtype != OSG_F16) continue;
            int sl = sole_consumer(op.m_output[0]);
            if (!is(sl, "osg.SiLU") || ops()[sl].m_input.size() != 1) continue;
            Operation f = op;
            f.m_attributes.emplace_back("osg_act", "silu");
            f.m_output = {ops()[sl].m_output[0]};
            dead[i] = 1;
            ops()[sl] = std::move(f);
        }
    }

    // ------------------------------------------------------------------------------------------------------------------
    // lowering
    // ------------------------------------------------------------------------------------------------------------------
    int in_val_raw(const Tensor& t) {   // may return a strided column view (Val::ld != 0)
        if (is_const_tensor(t)) return const_val(t);
        auto it = P.by_name.find(t.m_name);
        if (it == P.by_name.end()) throw std::invalid_argument("Model::get_tensor_data: input tensor not found: " + t.m_name);
        const int r = P.root_of(it->second);
        if (V(r).host_only && !V(r).dptr) {   // a plan-time value that a launch wants to read after all: give it device storage now
            const size_t bytes = std::max<size_t>(P.val_bytes(r), 8);
            V(r).dptr = P.small_alloc(bytes);
            if (V(r).dtype == OSG_I64) be.check(be.api.osg_upload_sync(be.ctx, V(r).dptr, V(r).host_i.data(), V(r).host_i.size() * 8), "osg_upload_sync");
            else if (V(r).dtype == OSG_F32 && P.fp16) {
                // the reference's fp32 results are rounded to fp16 when they are pushed with fp16 arithmetic on (push_tensor :3029-3034)
                std::vector<uint16_t> h(V(r).host_f.size());
                for (size_t k = 0; k < h.size(); k++) h[k] = float_to_half(V(r).host_f[k]);
                be.check(be.api.osg_upload_sync(be.ctx, V(r).dptr, h.data(), h.size() * 2), "osg_upload_sync");
                V(r).dtype = OSG_F16;
            } else if (V(r).dtype == OSG_F32) be.check(be.api.osg_upload_sync(be.ctx, V(r).dptr, V(r).host_f.data(), V(r).host_f.size() * 4), "osg_upload_sync");
            else throw std::invalid_argument("Model::get_tensor_data: unsupported tensor data format.");
        }
        return it->second;
    }
    // a constant, or an activation whose VALUE is known at plan time (the Shape -> Gather -> Concat -> Reshape chains exporters leave behind)
    const Val* hval(const Tensor& t) const {
        if (is_const_tensor(t)) return &P.vals[const_val(t)];
        auto it = P.by_name.find(t.m_name);
        if (it == P.by_name.end()) return nullptr;
        const Val& v = P.vals[it->second];
        return v.host_valid && v.host_only ? &v : nullptr;
    }
    int host_val(const Operation& op, const Shape& shape, std::vector<int64_t> ints, std::vector<float> floats = {}) {
        const bool is_int = floats.empty();
        check_out(op, shape);
        int v = P.new_val(op.m_output[0].m_name, shape, is_int ? OSG_I64 : OSG_F32, Lay::plain, false);
        V(v).is_const = true;
        V(v).host_valid = V(v).host_only = true;
        V(v).host_i = std::move(ints);
        V(v).host_f = std::move(floats);
        return v;
    }

    // Ops evaluated on the host while the plan is built: Shape (reference :7003), Range (:7589) and ConstantOfShape (:7543) always; Gather (:6316),
    // Cast (:7352), Concat (:4140), Unsqueeze / Squeeze / Reshape, Slice, Add / Sub / Mul / Div, Neg, Less / Greater / Equal / And (:7637), Where (:7034),
    // Expand (:7154) and Trilu when every operand is known at plan time and at least one of them is a plan-time VALUE (a shape chain, an int64
    // graph input, something computed from those) rather than a weight.  Tensor shapes and int64 inputs are fixed per plan (another shape or
    // another token re-plans), so the shape chains exporters leave behind and the mask / position subgraphs of the LLM graphs fold to
    // constants: int64 or fp32 host tensors that get device storage (fp32 -> f16 under fp16 arithmetic) only if a launch reads them.
    // Returns false when the op has to run on the device.
    struct HT {
        Shape shape;
        bool is_int = true;
        std::vector<double> v;
        long numel() const { return (long)v.size(); }
    };
    static HT to_ht(const Val* v) {
        HT h;
        h.shape = v->shape;
        h.is_int = v->dtype == OSG_I64;
        if (h.is_int) h.v.assign(v->host_i.begin(), v->host_i.end());
        else h.v.assign(v->host_f.begin(), v->host_f.end());
        return h;
    }
    int host_out(const Operation& op, HT h) {
        if (h.is_int) {
            std::vector<int64_t> o(h.v.size());
            for (size_t k = 0; k < o.size(); k++) o[k] = (int64_t)h.v[k];
            return host_val(op, h.shape, std::move(o));
        }
        std::vector<float> f(h.v.size());
        for (size_t k = 0; k < f.size(); k++) f[k] = (float)h.v[k];
        check_out(op, h.shape);
        int v = P.new_val(op.m_output[0].m_name, h.shape, OSG_F32, Lay::plain, false);
        V(v).is_const = true;
        V(v).host_valid = V(v).host_only = true;
        V(v).host_f = std::move(f);
        return v;
    }
    // numpy broadcasting of two host tensors through `fn`
    template <class F>
    HT broadcast2(const Operation& op, const HT& a, const HT& b, bool out_int, F&& fn) {
        const size_t r = std::max(a.shape.size(), b.shape.size());
        Shape as(r, 1), bs(r, 1), os(r, 1);
        std::copy(a.shape.begin(), a.shape.end(), as.begin() + (r - a.shape.size()));
        std::copy(b.shape.begin(), b.shape.end(), bs.begin() + (r - b.shape.size()));
        for (size_t k = 0; k < r; k++) {
            need(op, as[k] == bs[k] || as[k] == 1 || bs[k] == 1, "shapes of A and B not compatible.");
            os[k] = std::max(as[k], bs[k]);
        }
        HT o;
        o.shape = os;
        o.is_int = out_int;
        const long n = prod(os);
        need(op, n <= (1L << 24), "plan-time tensor too large.");
        o.v.resize((size_t)n);
        std::vector<long> idx(r, 0);
        for (long e = 0; e < n; e++) {
            long ia = 0, ib = 0;
            for (size_t k = 0; k < r; k++) {
                ia = ia * as[k] + (as[k] == 1 ? 0 : idx[k]);
                ib = ib * bs[k] + (bs[k] == 1 ? 0 : idx[k]);
            }
            o.v[(size_t)e] = fn(a.v[(size_t)ia], b.v[(size_t)ib]);
            for (long k = (long)r - 1; k >= 0; k--) {
                if (++idx[(size_t)k] < os[(size_t)k]) break;
                idx[(size_t)k] = 0;
            }
        }
        return o;
    }
    bool try_host_eval(const Operation& op) {
        const std::string& t = op.m_type;
        if (t == "Shape") {
            need(op, op.m_input.size() == 1, "wrong number of inputs.");
            need(op, op.m_output.size() == 1, "wrong number of outputs.");
            need(op, op.m_attributes.empty(), "unrecognized attribute (not implemented).");
            const int x = in_val_raw(op.m_input[0]);
            const Shape xs = V(x).shape;
            need(op, !xs.empty(), "shape of input not available.");
            host_val(op, {(long)xs.size()}, std::vector<int64_t>(xs.begin(), xs.end()));
            return true;
        }
        static const char* kOps[] = {"Gather", "Cast", "Concat", "Unsqueeze", "Squeeze", "Reshape", "Slice", "Add", "Sub", "Mul", "Div", "Neg", "Range",
                                     "ConstantOfShape", "Less", "Greater", "Equal", "And", "Where", "Expand", "Trilu"};
        bool known = false;
        for (auto* k : kOps) known |= t == k;
        if (!known || op.m_input.empty() || op.m_output.size() != 1) return false;
        std::vector<const Val*> in;
        bool any_value = false;
        for (auto& ti : op.m_input) {
            if (ti.m_name.empty()) { in.push_back(nullptr); continue; }
            const Val* v = hval(ti);
            if (!v || !v->host_valid || (v->dtype != OSG_I64 && v->dtype != OSG_F32 && v->host_f.empty())) return false;
            if (v->dtype != OSG_I64 && v->host_f.size() != (size_t)v->numel()) return false;   // (a weight without a host copy)
            any_value |= v->host_only;
            in.push_back(v);
        }
        // at least one operand must be a plan-time VALUE (not merely a small weight): all-weight ops stay on the device path
        if (!any_value || !in[0]) return false;
        // Gather / Slice / Concat / Expand / Unsqueeze / ... move DATA: the data operand itself has to be a plan-time value
        if ((t == "Gather" || t == "Slice" || t == "Unsqueeze" || t == "Squeeze" || t == "Reshape" || t == "Expand" || t == "Cast" || t == "Neg" || t == "Trilu") && !in[0]->host_only)
            return false;
        auto ints = [](const Val* v) { return v->dtype == OSG_I64; };
        auto attr_none = [&] { need(op, op.m_attributes.empty(), "unrecognized attribute (not implemented)."); };
        if (t == "Cast") {
            int to = -1;
            for (auto& a : op.m_attributes) {
                if (a.first == "to") to = std::stoi(a.second);
                else throw std::invalid_argument(op.m_type + ": unrecognized attribute (not implemented).");
            }
            need(op, to != -1, "'to' attribute not found.");
            HT x = to_ht(in[0]);
            if (to == 1) {
                need(op, x.is_int, "wrong data type of input (not implemented).");
                if (x.v.empty()) return false;
                x.is_int = false;
            } else if (to == 9 || to == 7 || to == 6) {
                for (auto& e : x.v) e = (double)(int64_t)e;
                x.is_int = true;
            } else
                throw std::invalid_argument(op.m_type + ": requested cast not implemented.");
            host_out(op, std::move(x));
            return true;
        }
        if (t == "Range") {
            need(op, op.m_input.size() == 3, "wrong number of inputs.");
            attr_none();
            for (int k = 0; k < 3; k++) need(op, in[k] && ints(in[k]) && in[k]->host_i.size() == 1 && in[k]->shape.empty(), "start, limit and delta must be int64 scalars (not implemented).");
            const int64_t st = in[0]->host_i[0], lim = in[1]->host_i[0], dl = in[2]->host_i[0];
            need(op, dl == 1, "delta must be 1 (not implemented).");
            need(op, st < lim, "start must be less than limit.");
            std::vector<int64_t> o;
            for (int64_t k = st; k < lim; k++) o.push_back(k);
            const long n = (long)o.size();
            host_val(op, {n}, std::move(o));
            return true;
        }
        if (t == "ConstantOfShape") {
            need(op, op.m_input.size() == 1, "wrong number of inputs.");
            std::string value;
            for (auto& a : op.m_attributes) {
                if (a.first == "value") value = a.second;
                else throw std::invalid_argument(op.m_type + ": unrecognized attribute (not implemented).");
            }
            need(op, !value.empty(), "'value' attribute not specified (not implemented).");
            need(op, in[0]->shape.size() == 1, "input must be 1D.");
            need(op, ints(in[0]), "wrong data type of input.");
            HT o;
            o.is_int = false;
            for (auto d : in[0]->host_i) o.shape.push_back((long)d);
            need(op, prod(o.shape) <= (1L << 24), "plan-time tensor too large.");
            o.v.assign((size_t)prod(o.shape), (double)std::stof(value));
            host_out(op, std::move(o));
            return true;
        }
        if (t == "Neg") {
            need(op, op.m_input.size() == 1, "wrong number of inputs.");
            HT x = to_ht(in[0]);
            for (auto& e : x.v) e = -e;
            host_out(op, std::move(x));
            return true;
        }
        if (t == "Less" || t == "Greater" || t == "Equal" || t == "And") {
            need(op, op.m_input.size() == 2 && in[1], "wrong number of inputs.");
            attr_none();
            const HT a = to_ht(in[0]), b = to_ht(in[1]);
            // two fp32 operands are compared as fixed point with 4 decimals (reference :7682-7683)
            const double fx = !a.is_int && !b.is_int ? 10000.0 : 1.0;
            const int kind = t == "Less" ? 0 : t == "Greater" ? 1 : t == "Equal" ? 2 : 3;
            host_out(op, broadcast2(op, a, b, true, [&](double x, double y) {
                         const int64_t p = a.is_int ? (int64_t)x : (int64_t)((float)x * (float)fx), q = b.is_int ? (int64_t)y : (int64_t)((float)y * (float)fx);
                         return (double)(kind == 0 ? p < q : kind == 1 ? p > q : kind == 2 ? p == q : (p && q));
                     }));
            return true;
        }
        if (t == "Where") {
            need(op, op.m_input.size() == 3 && in[1] && in[2], "wrong number of inputs.");
            attr_none();
            const HT c = to_ht(in[0]), a = to_ht(in[1]), b = to_ht(in[2]);
            need(op, !c.shape.empty(), "condition cannot be a scalar (not implemented).");
            need(op, c.is_int, "wrong data type of condition (not implemented).");
            need(op, (a.shape.empty() || a.shape == c.shape) && (b.shape.empty() || b.shape == c.shape), "shapes of condition, A and/or B not equal (broadcasting not implemented).");
            HT o;
            o.shape = c.shape;
            o.is_int = a.is_int || b.is_int;     // (an int64 operand makes the result int64, the other one is truncated: reference :7067-7110)
            o.v.resize(c.v.size());
            for (size_t k = 0; k < o.v.size(); k++) {
                double x = c.v[k] != 0.0 ? a.v[a.v.size() == 1 ? 0 : k] : b.v[b.v.size() == 1 ? 0 : k];
                o.v[k] = o.is_int ? (double)(int64_t)x : x;
            }
            host_out(op, std::move(o));
            return true;
        }
        if (t == "Expand") {
            need(op, op.m_input.size() == 2 && in[1], "wrong number of inputs.");
            attr_none();
            need(op, in[1]->shape.size() == 1 && ints(in[1]), "shape must be 1D.");
            HT ones;
            ones.is_int = true;
            for (auto d : in[1]->host_i) { need(op, d > 0, "dimension <= 0."); ones.shape.push_back((long)d); }
            need(op, prod(ones.shape) <= (1L << 24), "plan-time tensor too large.");
            ones.v.assign((size_t)prod(ones.shape), 1.0);
            const HT x = to_ht(in[0]);
            host_out(op, broadcast2(op, x, ones, x.is_int, [](double a, double) { return a; }));
            return true;
        }
        if (t == "Trilu") {
            need(op, op.m_input.size() == 2 && in[1], "wrong number of inputs.");
            for (auto& a : op.m_attributes) {
                if (a.first == "upper") need(op, a.second == "1", "'upper' must be 1 (not implemented).");
                else throw std::invalid_argument(op.m_type + ": unrecognized attribute (not implemented).");
            }
            HT x = to_ht(in[0]);
            need(op, !x.is_int, "wrong data type of input.");
            need(op, x.shape.size() == 2, "input must be 2D (not implemented).");
            need(op, ints(in[1]) && in[1]->shape.empty() && in[1]->host_i.size() == 1, "second input (k) must be a scalar (not implemented).");
            const long w = x.shape[1], h = x.shape[0], k = (long)in[1]->host_i[0];
            for (long y = 0; y < h; y++)
                for (long xx = 0; xx < w; xx++)
                    if (!(xx - k >= y)) x.v[(size_t)(y * w + xx)] = 0.0;
            host_out(op, std::move(x));
            return true;
        }
        if (t == "Gather") {
            need(op, op.m_input.size() == 2 && in[1], "wrong number of inputs.");
            int axis = 0;
            for (auto& a : op.m_attributes) {
                if (a.first == "axis") axis = std::stoi(a.second);
                else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
            }
            need(op, ints(in[1]), "wrong data type of indices.");
            need(op, in[0]->shape.size() == 1 && (axis == 0 || axis == -1), "axis must be 0 (not implemented).");
            const HT x = to_ht(in[0]);
            HT o;
            o.is_int = x.is_int;
            o.shape = in[1]->shape;                       // 0-d indices -> 0-d output, 1-d -> 1-d (reference :6391-6394)
            for (int64_t i : in[1]->host_i) {
                if (i < 0) i += (int64_t)x.v.size();
                need(op, i >= 0 && i < (int64_t)x.v.size(), "invalid index in indices.");
                o.v.push_back(x.v[(size_t)i]);
            }
            host_out(op, std::move(o));
            return true;
        }
        if (t == "Concat") {
            HT o;
            o.is_int = true;
            for (auto* v : in) {
                need(op, v && v->shape.size() <= 1, "invalid shape of inputs.");
                const HT x = to_ht(v);
                o.is_int &= x.is_int;
                o.v.insert(o.v.end(), x.v.begin(), x.v.end());
            }
            o.shape = {(long)o.v.size()};
            host_out(op, std::move(o));
            return true;
        }
        if (t == "Unsqueeze" || t == "Squeeze") {
            HT x = to_ht(in[0]);
            std::vector<long> axes;
            if (op.m_input.size() > 1 && in[1]) {
                need(op, ints(in[1]), "wrong data type of axes.");
                axes.assign(in[1]->host_i.begin(), in[1]->host_i.end());
            } else if (auto* a = attr(op, "axes"))
                for (int k : int_list(*a)) axes.push_back(k);
            need(op, !axes.empty(), "axes cannot be empty (not implemented).");
            const long rank_out = t == "Unsqueeze" ? (long)x.shape.size() + (long)axes.size() : (long)x.shape.size();
            for (auto& a : axes) {
                if (a < 0) a += rank_out;
                need(op, a >= 0 && a < rank_out, "wrong data in axes.");
            }
            std::sort(axes.begin(), axes.end());
            if (t == "Unsqueeze")
                for (long a : axes) x.shape.insert(x.shape.begin() + a, 1);
            else
                for (auto it = axes.rbegin(); it != axes.rend(); ++it) {
                    need(op, x.shape[(size_t)*it] == 1, "wrong data in axes.");
                    x.shape.erase(x.shape.begin() + *it);
                }
            host_out(op, std::move(x));
            return true;
        }
        if (t == "Reshape") {
            need(op, op.m_input.size() == 2 && in[1] && ints(in[1]), "wrong data type of shape.");
            HT x = to_ht(in[0]);
            Shape os;
            long unknown = -1, known_n = 1;
            for (size_t k = 0; k < in[1]->host_i.size(); k++) {
                long dsz = (long)in[1]->host_i[k];
                if (dsz == 0) { need(op, k < x.shape.size(), "invalid shape."); dsz = x.shape[k]; }
                if (dsz == -1) { need(op, unknown < 0, "invalid shape."); unknown = (long)k; os.push_back(1); continue; }
                os.push_back(dsz);
                known_n *= dsz;
            }
            if (unknown >= 0) { need(op, known_n > 0 && x.numel() % known_n == 0, "invalid shape."); os[(size_t)unknown] = x.numel() / known_n; }
            need(op, prod(os) == x.numel(), "invalid shape.");
            x.shape = os;
            host_out(op, std::move(x));
            return true;
        }
        if (t == "Slice") {
            need(op, op.m_input.size() >= 3 && in[1] && in[2] && in[0]->shape.size() == 1 && in[1]->host_i.size() == 1 && in[2]->host_i.size() == 1, "unsupported slice of a plan-time vector (not implemented).");
            if (op.m_input.size() > 4 && in[4]) need(op, in[4]->host_i.size() == 1 && in[4]->host_i[0] == 1, "unsupported steps value(s) (not implemented).");
            HT x = to_ht(in[0]);
            int64_t b = in[1]->host_i[0], e = in[2]->host_i[0];
            const int64_t n = (int64_t)x.v.size();
            if (b < 0) b += n;
            if (e < 0) e += n;
            b = std::min(std::max<int64_t>(b, 0), n);
            e = std::min(std::max<int64_t>(e, 0), n);
            need(op, b < e, "invalid value(s) in starts and/or ends.");
            HT o;
            o.is_int = x.is_int;
            o.shape = {(long)(e - b)};
            o.v.assign(x.v.begin() + b, x.v.begin() + e);
            host_out(op, std::move(o));
            return true;
        }
        // Add / Sub / Mul / Div: int64 when both operands are, else fp32 (numpy broadcasting)
        need(op, op.m_input.size() == 2 && in[1], "wrong number of inputs.");
        const HT a = to_ht(in[0]), b = to_ht(in[1]);
        const bool oi = a.is_int && b.is_int;
        const int kind = t == "Add" ? 0 : t == "Sub" ? 1 : t == "Mul" ? 2 : 3;
        host_out(op, broadcast2(op, a, b, oi, [&](double x, double y) -> double {
                     if (oi) {
                         const int64_t p = (int64_t)x, q = (int64_t)y;
                         if (kind == 3) need(op, q != 0, "division by zero.");
                         return (double)(kind == 0 ? p + q : kind == 1 ? p - q : kind == 2 ? p * q : p / q);
                     }
                     const float p = (float)x, q = (float)y;
                     return (double)(kind == 0 ? p + q : kind == 1 ? p - q : kind == 2 ? p * q : p / q);
                 }));
        return true;
    }

    // ==================================================================================================================
    // uint8 arithmetic (m_use_uint8_arithmetic; the reference's W8A8 path -- in practice the VAE decoder of `sd --rpi-lowmem`,
    // src/sd.cpp:1212-1222).  One launch per graph op: every op re-quantises to its OWN (scale, zero point) -- derived from
    // range_data.txt by Model::range_to_scale exactly where the reference derives them -- so nothing may be fused without changing
    // codes.  Quantisation parameters are read from the vals at RUN time (a pushed input is quantised per run, reference :3024-3028).
    // ==================================================================================================================
    qu8::QParams out_q(const Operation& op) {
        auto it = m.m_range_data.find(op.m_name);
        if (it == m.m_range_data.end()) throw std::invalid_argument(op.m_type + ": range data not found.");
        return qu8::range_to_scale(it->second.first, it->second.second);
    }
    int out_val_u8(const Operation& op, const Shape& shape, Lay lay, bool batched, qu8::QParams q) {
        int y = out_val(op, shape, lay, batched, OSG_U8);
        V(y).qscale = q.scale;
        V(y).qzp = (int)q.zero_point;
        return y;
    }
    void need_u8(const Operation& op, int v, const char* what) {
        if (V(v).dtype != OSG_U8) throw std::invalid_argument(op.m_type + ": wrong data type of " + what + ".");
    }
    // a 256-byte device table that lives as long as the plan
    void* lut_alloc(size_t bytes) {
        return P.small_alloc(bytes);
    }

    void lower_u8(const Operation& op) {
        const std::string& t = op.m_type;
        if (t == "Conv") return lower_conv_u8(op);
        if (t == "MatMul") return lower_matmul_u8(op);
        if (t == "Add" || t == "Mul") return lower_binary_u8(op);
        if (t == "Sigmoid") return lower_sigmoid_u8(op);
        if (t == "InstanceNormalization") return lower_instance_norm_u8(op);
        if (t == "osg.qu8.InstanceNormNHWC") return lower_instance_norm_u8_nhwc(op);
        if (t == "osg.qu8.AffineAct" || t == "osg.qu8.NormAffineAct") return lower_affine_act_u8(op);
        if (t == "Softmax") return lower_softmax_u8(op);
        if (t == "Reshape" || t == "Flatten" || t == "Unsqueeze" || t == "Squeeze" || t == "Transpose" || t == "Resize") {
            // the codes are re-arranged, scale and zero point carried over (reference :4783, :5231, :6251)
            const int x = in_val_raw(op.m_input[0]);
            if (t == "Reshape") lower_reshape(op);
            else if (t == "Flatten") lower_flatten(op);
            else if (t == "Unsqueeze" || t == "Squeeze") lower_squeeze(op, t == "Unsqueeze");
            else if (t == "Transpose") lower_transpose(op);
            else lower_resize(op);
            P.share_q(P.by_name.at(op.m_output[0].m_name), x);
            return;
        }
        throw std::invalid_argument("Model::run: operation not implemented with uint8 arithmetic on the HIP backend: " + t);
    }

    // Conv, uint8 branch (reference :4629-4690 -> XnnPack::convolution<uint8_t,int32_t> :1292)
    void lower_conv_u8(const Operation& op) {
        need(op, op.m_input.size() == 2 || op.m_input.size() == 3, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        std::vector<int> dil = {1, 1}, ks, pads = {0, 0, 0, 0}, strides = {1, 1};
        int group = 1;
        for (auto& a : op.m_attributes) {
            if (a.first == "dilations") dil = int_list(a.second);
            else if (a.first == "group") group = std::stoi(a.second);
            else if (a.first == "kernel_shape") ks = int_list(a.second);
            else if (a.first == "pads") pads = int_list(a.second);
            else if (a.first == "strides") strides = int_list(a.second);
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        int x = in_val(op.m_input[0]);
        need(op, V(x).shape.size() == 4, "Conv1D / non 4-D input not implemented on the HIP backend.");
        for (int d : dil) need(op, d == 1, "dilations != 1 not supported (not implemented).");
        need(op, group == 1, "group != 1 not supported (not implemented).");
        need(op, pads.size() == 4 && strides.size() == 2, "invalid pads/strides.");
        need_u8(op, x, "X");
        x = P.ensure_nhwc(x);
        const int w = in_val(op.m_input[1]);
        need_u8(op, w, "W");
        const Shape ws = V(w).shape, xs = V(x).shape;
        need(op, V(w).is_const && V(w).lay == Lay::nhwc && ws.size() == 4, "weights must be a static *_nchw.bin tensor.");
        need(op, xs[0] == 1 && N == 1, "uint8 arithmetic runs one sample per pass (every pushed sample has its own scale).");
        const long Cin = xs[1], H = xs[2], W = xs[3], Cout = ws[0], KH = ws[2], KW = ws[3];
        need(op, ws[1] == Cin, "invalid shape of weights.");
        if (!ks.empty()) need(op, ks.size() == 2 && ks[0] == KH && ks[1] == KW, "kernel_shape does not match the weights.");
        const int ph = pads[0] + pads[2], pw = pads[1] + pads[3];
        const int pt = ph / 2, pb = ph - pt, pl = pw / 2, pr = pw - pl;
        const long Ho = (H + ph - KH) / strides[0] + 1, Wo = (W + pw - KW) / strides[1] + 1;
        int bias = -1;
        if (op.m_input.size() == 3 && !op.m_input[2].m_name.empty()) {
            bias = in_val(op.m_input[2]);
            need(op, V(bias).numel() == Cout && V(bias).dtype == OSG_F32, "wrong data type of B.");
        }
        const qu8::QParams oq = out_q(op);
        const int y = out_val_u8(op, {1, Cout, Ho, Wo}, Lay::nhwc, V(x).batched, oq);
        const int sh = strides[0], sw = strides[1];
        std::vector<int> reads = {x, w};
        if (bias >= 0) reads.push_back(bias);
        // the pipelined kernel's table of code sums per filter tap (include/osgpu.h osg_qu8_conv2d_nhwc_t): a function of the weight alone, so a weight
        // that stays at its address gets it once, here (kept with the Model's constants under the weight's name); a weight that travels through the
        // streaming ring (VRAM budget) has none -- the library then rebuilds it in its workspace before every launch
        int* taps = nullptr;
        if (V(w).dptr && (pt || pl || pb || pr) && KH * KW <= 32) {
            bool fresh;
            taps = (int*)P.const_alloc(V(w).name + "|q8taps", (size_t)Cout * KH * KW * sizeof(int), &fresh);
            if (fresh) be.check(be.api.osg_qu8_conv_tap_sums(be.ctx, V(w).dptr, (int)Cout, (int)KH, (int)KW, (int)Cin, taps), "osg_qu8_conv_tap_sums");
        }
        P.add_step("Conv qu8 " + op.m_name, reads, {y}, [=, this] {
            const Val& qx = P.qv(x);
            be.check(be.api.osg_qu8_conv2d_nhwc_t(be.ctx, P.ptr(x), qx.qscale, qx.qzp, P.ptr(w), P.vals[w].qscale, P.vals[w].qzp,
                                                  bias >= 0 ? (const float*)P.ptr(bias) : nullptr, oq.scale, (int)oq.zero_point, P.ptr(y), 1, (int)H, (int)W, (int)Cin,
                                                  (int)Cout, (int)KH, (int)KW, sh, sw, pt, pl, pb, pr, taps),
                     "Conv");
        });
        P.steps.back().flops = 2.0 * Ho * Wo * Cout * KH * KW * Cin;
    }

    // MatMul, uint8 branch (reference :5779-5837 -> XnnPack::matrix_multiply<uint8_t> :1035): [.., M,K] x [K,N] weight or batched [n,M,K] x [n,K,N]
    void lower_matmul_u8(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        need(op, N == 1, "uint8 arithmetic runs one sample per pass (every pushed sample has its own scale).");
        int a = P.ensure_plain(in_val(op.m_input[0])), b = in_val(op.m_input[1]);
        need_u8(op, a, "input");
        need_u8(op, b, "input");
        const qu8::QParams oq = out_q(op);
        if (V(b).is_const && V(b).shape.size() == 2) {
            const Shape as = V(a).shape;
            const long K = V(b).shape[0], Nn = V(b).shape[1];
            need(op, !as.empty() && as.back() == K, "invalid shape of inputs.");
            const int wq = weight_nk_u8(b);
            Shape os = as;
            os.back() = Nn;
            const int y = out_val_u8(op, os, Lay::plain, V(a).batched, oq);
            const long M = prod(as) / K;
            P.add_step("MatMul qu8 " + op.m_name, {a, wq}, {y}, [=, this] {
                const Val& qa = P.qv(a);
                be.check(be.api.osg_qu8_gemm(be.ctx, P.ptr(a), K, qa.qscale, qa.qzp, P.ptr(wq), P.vals[wq].qscale, P.vals[wq].qzp, nullptr, oq.scale,
                                             (int)oq.zero_point, P.ptr(y), (int)M, (int)Nn, (int)K, 1, 0, 0, 0),
                         "MatMul");
            });
            P.steps.back().flops = 2.0 * M * Nn * K;
            return;
        }
        b = P.ensure_plain(b);
        Shape as = V(a).shape, bs = V(b).shape;
        bool lead1 = false;
        if (as.size() == 4 && as[0] == 1) { as.erase(as.begin()); lead1 = true; }
        if (bs.size() == 4 && bs[0] == 1) bs.erase(bs.begin());
        need(op, as.size() == 3 && bs.size() == 3 && as[0] == bs[0] && as[2] == bs[1], "invalid shape of inputs.");
        const long n = as[0], M = as[1], K = as[2], Nn = bs[2];
        Shape os = {n, M, Nn};
        if (lead1) os.insert(os.begin(), 1);
        const int y = out_val_u8(op, os, Lay::plain, false, oq);
        // the [K,N] operand is an activation: re-laid out to K-contiguous [N,K] by a transpose launch of its own
        const int bt = P.new_val("", {n, Nn, K}, OSG_U8, Lay::plain, false);
        P.add_step("MatMul qu8/T " + op.m_name, {b}, {bt}, [=, this] {
            long sh[3] = {n, K, Nn};
            int pm[3] = {0, 2, 1};
            be.check(be.api.osg_transpose(be.ctx, 1, P.ptr(b), P.ptr(bt), 3, sh, pm), "MatMul");
        });
        P.add_step("MatMul qu8 " + op.m_name, {a, bt}, {y}, [=, this] {
            const Val &qa = P.qv(a), &qb = P.qv(b);
            be.check(be.api.osg_qu8_gemm(be.ctx, P.ptr(a), K, qa.qscale, qa.qzp, P.ptr(bt), qb.qscale, qb.qzp, nullptr, oq.scale, (int)oq.zero_point, P.ptr(y),
                                         (int)M, (int)Nn, (int)K, (int)n, M * K, Nn * K, M * Nn),
                     "MatMul");
        });
        P.steps.back().flops = 2.0 * n * M * Nn * K;
    }

    // Add / Mul, uint8 branches (reference :5105-5124, :3977-3996 -> XnnPack::add / multiply with quint8 parameters)
    void lower_binary_u8(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        int a = in_val(op.m_input[0]), b = in_val(op.m_input[1]);
        need_u8(op, a, "inputs");
        need_u8(op, b, "inputs");
        const osg_binary_kind kind = op.m_type == "Add" ? OSG_BIN_ADD : OSG_BIN_MUL;
        const Shape as = V(a).shape, bs = V(b).shape;
        const size_t rank = std::max(as.size(), bs.size());
        Shape os(rank);
        for (size_t i = 0; i < rank; i++) {
            long da = i + as.size() >= rank ? as[i + as.size() - rank] : 1;
            long db = i + bs.size() >= rank ? bs[i + bs.size() - rank] : 1;
            need(op, da == db || da == 1 || db == 1, "shapes are not broadcastable.");
            os[i] = std::max(da, db);
        }
        auto per_channel = [&](int v, long C) {
            const Shape& s = V(v).shape;
            if (V(v).lay == Lay::nhwc) return false;
            if (V(v).numel() == 1) return true;
            if (V(v).numel() != C) return false;
            return (s.size() == 3 && s[0] == C) || (s.size() == 4 && s[1] == C);
        };
        Lay olay = Lay::plain;
        Shape pa, pb;
        if (V(a).lay == Lay::nhwc || V(b).lay == Lay::nhwc) {
            int x = V(a).lay == Lay::nhwc ? a : b, o = x == a ? b : a;
            const Shape xs = V(x).shape;
            const long C = xs[1], HW = xs[2] * xs[3];
            if (V(o).shape == xs && xs.size() == 4) {
                // same-shape operands: stay channels-last (the convolutions either side want it); a plain operand is transposed ONCE here instead of
                // the NHWC one being transposed now and the result transposed back in front of the next convolution
                if (V(o).lay != Lay::nhwc) {
                    const int on = P.ensure_nhwc(o);
                    (o == a ? a : b) = on;
                }
                olay = Lay::nhwc;
                pa = pb = {HW, C};
            } else if (per_channel(o, C) && os == xs) {
                olay = Lay::nhwc;
                Shape px = {HW, C}, po = {1, V(o).numel() == 1 ? 1 : C};
                pa = x == a ? px : po;
                pb = x == a ? po : px;
            } else {
                a = P.ensure_plain(a);
                b = P.ensure_plain(b);
            }
        }
        if (olay == Lay::plain) { pa = V(a).shape; pb = V(b).shape; }
        const qu8::QParams oq = out_q(op);
        const int y = out_val_u8(op, os, olay, V(a).batched || V(b).batched, oq);
        const size_t prank = std::max(pa.size(), pb.size());
        need(op, prank >= 1 || true, "");
        const size_t pr = std::max<size_t>(prank, 1);
        need(op, pr <= 6, "rank too large for the device broadcast kernel.");
        std::vector<long> sa(pr, 1), sb(pr, 1);
        for (size_t i = 0; i < pa.size(); i++) sa[pr - pa.size() + i] = pa[i];
        for (size_t i = 0; i < pb.size(); i++) sb[pr - pb.size() + i] = pb[i];
        P.add_step(op.m_type + " qu8 " + op.m_name, {a, b}, {y}, [=, this] {
            const Val &qa = P.qv(a), &qb = P.qv(b);
            be.check(be.api.osg_qu8_binary(be.ctx, kind, P.ptr(a), sa.data(), qa.qscale, qa.qzp, P.ptr(b), sb.data(), qb.qscale, qb.qzp, P.ptr(y), oq.scale,
                                           (int)oq.zero_point, (int)pr),
                     op.m_type.c_str());
        });
    }

    // Sigmoid, uint8 branch (reference :4412-4481): a function of the input code -> 256-entry table built on the host with the host's expf
    void lower_sigmoid_u8(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        const int x = in_val(op.m_input[0]);
        need_u8(op, x, "input");
        const qu8::QParams oq = out_q(op);
        const int y = out_val_u8(op, V(x).shape, V(x).lay, V(x).batched, oq);
        const long n = P.total_elems(x);
        void* lut = lut_alloc(256);
        const bool dyn = P.qv(x).qdyn;
        auto build = [=, this] {
            const Val& qx = P.qv(x);
            uint8_t t[256];
            qu8::sigmoid_lut(qu8::QParams{qx.qscale, (uint8_t)qx.qzp}, oq, t);
            be.check(be.api.osg_upload_sync(be.ctx, lut, t, 256), "osg_upload_sync");
        };
        if (!dyn) build();
        P.add_step("Sigmoid qu8 " + op.m_name, {x}, {y}, [=, this] {
            if (dyn) build();      // the input scale changes from run to run: rebuild (256 expf) before the lookup
            be.check(be.api.osg_qu8_lut(be.ctx, P.ptr(x), P.ptr(y), n, lut), "Sigmoid");
        });
    }

    // InstanceNormalization, uint8 branch (reference :4987-5043): input [1,G,L]
    void lower_instance_norm_u8(const Operation& op) {
        need(op, op.m_input.size() == 3, "wrong number of inputs.");
        const int x = P.ensure_plain(in_val(op.m_input[0]));
        need_u8(op, x, "input");
        const int sc = in_val(op.m_input[1]), bi = in_val(op.m_input[2]);
        const Shape s = V(x).shape;
        need(op, s.size() == 3 && s[0] == 1, "input shape must be [1,G,L] (not implemented).");
        need(op, V(sc).numel() == s[1] && V(bi).numel() == s[1] && V(sc).dtype == OSG_F32 && V(bi).dtype == OSG_F32, "invalid scale/bias.");
        float eps = 1e-5f;
        for (auto& a : op.m_attributes) {
            if (a.first == "epsilon") eps = std::stof(a.second);
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        const qu8::QParams oq = out_q(op);
        const int y = out_val_u8(op, s, Lay::plain, V(x).batched, oq);
        const long rows = s[1], L = s[2];
        P.add_step("InstanceNorm qu8 " + op.m_name, {x, sc, bi}, {y}, [=, this] {
            const Val& qx = P.qv(x);
            be.check(be.api.osg_qu8_instance_norm(be.ctx, P.ptr(x), P.ptr(y), (int)rows, L, (int)rows, (const float*)P.ptr(sc), (const float*)P.ptr(bi), eps, qx.qscale,
                                                  qx.qzp, oq.scale, (int)oq.zero_point),
                     "InstanceNormalization");
        });
    }

    qu8::QParams range_q(const Operation& op, const std::string& name) {
        auto it = m.m_range_data.find(name);
        if (it == m.m_range_data.end()) throw std::invalid_argument(op.m_type + ": range data not found.");
        return qu8::range_to_scale(it->second.first, it->second.second);
    }
    void lower_affine_act_u8(const Operation& op) {
        const bool normed = op.m_type == "osg.qu8.NormAffineAct";
        int x = in_val(op.m_input[0]);
        const int g = in_val(op.m_input[1]), b = in_val(op.m_input[2]);
        need_u8(op, x, "inputs");
        need_u8(op, g, "inputs");
        need_u8(op, b, "inputs");
        const Shape s = V(x).shape;
        need(op, s.size() == 4 && s[0] == 1, "input shape must be [1,C,H,W] (not implemented).");
        const long C = s[1], HW = s[2] * s[3], n = P.total_elems(x);
        need(op, V(g).numel() == C && V(b).numel() == C, "invalid shape of the per-channel operands.");
        const qu8::QParams qm = range_q(op, *attr(op, "mul")), qa = range_q(op, *attr(op, "add"));
        const bool silu = attr(op, "sigmoid") != nullptr;
        qu8::QParams qs{}, qo = qa;
        void* lut = nullptr;
        if (silu) {
            qs = range_q(op, *attr(op, "sigmoid"));
            qo = range_q(op, *attr(op, "mul2"));
            lut = lut_alloc(256);
            uint8_t t[256];
            qu8::sigmoid_lut(qa, qs, t);          // (the Sigmoid's input is the Add's output: parameters from range data, never dynamic)
            be.check(be.api.osg_upload_sync(be.ctx, lut, t, 256), "osg_upload_sync");
        }
        int sc = -1, bi = -1;
        float eps = 1e-5f;
        long G = 0;
        qu8::QParams qn{};
        if (normed) {
            sc = in_val(op.m_input[3]), bi = in_val(op.m_input[4]);
            if (auto* e = attr(op, "norm_epsilon")) eps = std::stof(*e);
            if (auto* gr = attr(op, "norm_groups")) G = std::stol(*gr);
            need(op, G > 0 && C % G == 0 && V(sc).numel() == G && V(bi).numel() == G && V(sc).dtype == OSG_F32 && V(bi).dtype == OSG_F32, "invalid scale/bias.");
            qn = range_q(op, *attr(op, "norm"));
            if (V(x).lay != Lay::nhwc) {
                // the producer left it in the logical layout: the [G][L] row kernels as they are, then the affine pass below on their output
                const int xi = P.ensure_plain(x), nv = P.new_val("", s, OSG_U8, Lay::plain, V(x).batched);
                V(nv).qscale = qn.scale;
                V(nv).qzp = (int)qn.zero_point;
                const long L = (C / G) * HW;
                P.add_step("InstanceNorm qu8 " + *attr(op, "norm"), {xi, sc, bi}, {nv}, [=, this] {
                    const Val& qx = P.qv(xi);
                    be.check(be.api.osg_qu8_instance_norm(be.ctx, P.ptr(xi), P.ptr(nv), (int)G, L, (int)G, (const float*)P.ptr(sc), (const float*)P.ptr(bi), eps, qx.qscale, qx.qzp,
                                                          qn.scale, (int)qn.zero_point),
                             "InstanceNormalization");
                });
                x = nv;
            }
        }
        if (normed && V(x).lay == Lay::nhwc) {
            const int y = out_val_u8(op, s, Lay::nhwc, V(x).batched, qo);
            P.add_step("NormAffineAct qu8 " + op.m_name, {x, g, b, sc, bi}, {y}, [=, this] {
                const Val &qx = P.qv(x), &qg = P.qv(g), &qb = P.qv(b);
                be.check(be.api.osg_qu8_norm_affine_act_nhwc(be.ctx, P.ptr(x), HW, (int)C, (int)G, (int)G, (const float*)P.ptr(sc), (const float*)P.ptr(bi), eps, qx.qscale, qx.qzp,
                                                             qn.scale, (int)qn.zero_point, P.ptr(g), qg.qscale, qg.qzp, qm.scale, (int)qm.zero_point, P.ptr(b), qb.qscale, qb.qzp,
                                                             qa.scale, (int)qa.zero_point, lut, qs.scale, (int)qs.zero_point, qo.scale, (int)qo.zero_point, P.ptr(y)),
                         "NormAffineAct");
            });
            return;
        }
        const int y = out_val_u8(op, s, V(x).lay, V(x).batched, qo);
        const long inner = V(x).lay == Lay::nhwc ? 1 : HW;
        P.add_step("AffineAct qu8 " + op.m_name, {x, g, b}, {y}, [=, this] {
            const Val &qx = P.qv(x), &qg = P.qv(g), &qb = P.qv(b);
            be.check(be.api.osg_qu8_affine_act(be.ctx, P.ptr(x), qx.qscale, qx.qzp, P.ptr(g), qg.qscale, qg.qzp, qm.scale, (int)qm.zero_point, P.ptr(b), qb.qscale, qb.qzp,
                                               qa.scale, (int)qa.zero_point, lut, qs.scale, (int)qs.zero_point, qo.scale, (int)qo.zero_point, P.ptr(y), n, (int)C, inner),
                     "AffineAct");
        });
    }

    void lower_instance_norm_u8_nhwc(const Operation& op) {
        const int x0 = in_val(op.m_input[0]);
        need_u8(op, x0, "input");
        const int sc = in_val(op.m_input[1]), bi = in_val(op.m_input[2]);
        const Shape s = V(x0).shape;
        need(op, s.size() == 4 && s[0] == 1, "input shape must be [1,C,H,W] (not implemented).");
        float eps = 1e-5f;
        long G = 0;
        for (auto& a : op.m_attributes) {
            if (a.first == "epsilon") eps = std::stof(a.second);
            else if (a.first == "groups") G = std::stol(a.second);
            else throw std::invalid_argument("InstanceNormalization: unrecognized attribute: " + a.first + ".");
        }
        const long C = s[1], HW = s[2] * s[3];
        need(op, G > 0 && C % G == 0, "invalid number of groups.");
        need(op, V(sc).numel() == G && V(bi).numel() == G && V(sc).dtype == OSG_F32 && V(bi).dtype == OSG_F32, "invalid scale/bias.");
        const qu8::QParams oq = out_q(op);
        if (V(x0).lay != Lay::nhwc) {
            // the producer left it in the logical layout: the plain [G][L] kernels apply as they are (rows are contiguous there)
            const int x = P.ensure_plain(x0);
            const int y = out_val_u8(op, s, Lay::plain, V(x).batched, oq);
            const long L = (C / G) * HW;
            P.add_step("InstanceNorm qu8 " + op.m_name, {x, sc, bi}, {y}, [=, this] {
                const Val& qx = P.qv(x);
                be.check(be.api.osg_qu8_instance_norm(be.ctx, P.ptr(x), P.ptr(y), (int)G, L, (int)G, (const float*)P.ptr(sc), (const float*)P.ptr(bi), eps, qx.qscale, qx.qzp,
                                                      oq.scale, (int)oq.zero_point),
                         "InstanceNormalization");
            });
            return;
        }
        const int x = x0;
        const int y = out_val_u8(op, s, Lay::nhwc, V(x).batched, oq);
        P.add_step("InstanceNorm qu8 nhwc " + op.m_name, {x, sc, bi}, {y}, [=, this] {
            const Val& qx = P.qv(x);
            be.check(be.api.osg_qu8_instance_norm_nhwc(be.ctx, P.ptr(x), P.ptr(y), HW, (int)C, (int)G, (int)G, (const float*)P.ptr(sc), (const float*)P.ptr(bi), eps, qx.qscale,
                                                       qx.qzp, oq.scale, (int)oq.zero_point),
                     "InstanceNormalization");
        });
    }

    // Softmax, uint8 branch (reference :5960-5975 -> XnnPack::softmax<uint8_t>): last axis; output scale 1/256, zero point 0
    void lower_softmax_u8(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        const int x = P.ensure_plain(in_val(op.m_input[0]));
        need_u8(op, x, "input");
        const Shape s = V(x).shape;
        int axis = -1;
        for (auto& a : op.m_attributes) {
            if (a.first == "axis") axis = std::stoi(a.second);
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        if (axis < 0) axis += (int)s.size();
        need(op, axis == (int)s.size() - 1, "uint8 softmax over a non-last axis is not implemented on the HIP backend.");
        const int y = out_val_u8(op, s, Lay::plain, V(x).batched, qu8::QParams{0x1.0p-8f, 0});
        const long C = s.back(), rows = P.total_elems(x) / C;
        void* lut = lut_alloc(256 * 4);
        const bool dyn = P.qv(x).qdyn;
        auto build = [=, this] {
            uint32_t t[256];
            qu8::softmax_lut(P.qv(x).qscale, (size_t)C, t);
            be.check(be.api.osg_upload_sync(be.ctx, lut, t, sizeof t), "osg_upload_sync");
        };
        if (!dyn) build();
        P.add_step("Softmax qu8 " + op.m_name, {x}, {y}, [=, this] {
            if (dyn) build();
            be.check(be.api.osg_qu8_softmax_last(be.ctx, P.ptr(x), P.ptr(y), rows, C, lut), "Softmax");
        });
    }

    // Gather on the device (reference :6316-6498): rows of a [rows, els] view along axis 0 (leading 1-dims stripped), static int64 indices
    void lower_gather(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        int axis = 0;
        for (auto& a : op.m_attributes) {
            if (a.first == "axis") axis = std::stoi(a.second);
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        const Val* iv = hval(op.m_input[1]);
        need(op, iv && iv->dtype == OSG_I64 && iv->host_valid, "wrong data type of indices.");
        int x = P.ensure_plain(in_val(op.m_input[0]));
        Shape xs = V(x).shape;
        if (axis < 0) axis += (int)xs.size();
        const int prev_axis = axis;
        while (axis > 0 && !xs.empty() && xs[0] == 1) { xs.erase(xs.begin()); axis--; }
        need(op, axis == 0, "axis must be 0 (not implemented).");
        const bool in1d = xs.size() == 1;
        if (in1d) xs.insert(xs.begin(), 1);
        Shape is = iv->shape;
        const bool idx0d = is.empty();
        if (idx0d) is.push_back(1);
        const bool idx1d = is.size() == 1;
        if (idx1d) is.insert(is.begin(), 1);
        need(op, is.size() == 2 && is[0] == 1, "shape of indices must be (1,D) (not implemented).");
        Shape over;
        if (xs.size() > 2) {
            over.assign(xs.begin() + 1, xs.end());
            xs = {xs[0], prod(over)};
        }
        need(op, xs.size() == 2, "input must be 2D or more.");
        Shape os = {1, is[1], xs[1]};
        if (in1d && idx0d) os.clear();
        else if (idx1d) os.erase(os.begin());
        for (int i = 0; i < prev_axis; i++) os.insert(os.begin(), 1);
        if (!over.empty()) os = over;
        const long dim = in1d ? xs[1] : xs[0], els = in1d ? 1 : xs[1];
        std::vector<int64_t> idx = iv->host_i;
        for (auto& i : idx) {
            if (i < 0) i += dim;
            need(op, i >= 0 && i < dim, "invalid index in indices.");
        }
        const int idev = P.new_val("", {(long)idx.size()}, OSG_I64, Lay::plain, false);
        V(idev).is_const = true;
        bool fresh;
        std::string tag = "gather|" + op.m_name;
        V(idev).dptr = P.const_alloc(tag, std::max<size_t>(idx.size() * 8, 8), &fresh);
        be.check(be.api.osg_upload_sync(be.ctx, V(idev).dptr, idx.data(), idx.size() * 8), "osg_upload_sync");
        int y = out_val(op, os, Lay::plain, V(x).batched, V(x).dtype);
        const int es = (int)esize(V(x).dtype);
        const long nb = B(x), n_idx = (long)idx.size(), per_in = V(x).numel(), per_out = n_idx * els;
        P.add_step("Gather " + op.m_name, {x, idev}, {y}, [=, this] {
            for (long b = 0; b < nb; b++)
                be.check(be.api.osg_gather_rows(be.ctx, es, (const char*)P.ptr(x) + b * per_in * es, (const int64_t*)P.ptr(idev), (char*)P.ptr(y) + b * per_out * es, n_idx,
                                                els, dim),
                         "Gather");
        });
    }

    // Cast on the device (reference :7352-7423): the device keeps f16 activations only, so FLOAT -> FLOAT16 / FLOAT are relabelings there
    void lower_cast(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        throw std::invalid_argument(op.m_type + ": requested cast not implemented (only casts of plan-time integer values are supported on the HIP backend).");
    }

    // m_requires_upcast (reference get_tensor_data :2847-2848: the operands of a flagged op are handed over as fp32, the op runs in fp32; its fp32
    // result stays fp32 only while the NEXT op is its sole consumer, push_tensor :3008-3034): cur_up is set while a flagged op is lowered
    bool cur_up = false;
    std::map<int, int> cast_cache32, cast_cache16;
    int cast_val(int v, osg_dtype to) {
        auto& cache = to == OSG_F32 ? cast_cache32 : cast_cache16;
        auto it = cache.find(v);
        if (it != cache.end()) return it->second;
        const int src = P.ensure_plain(v);
        const int y = P.new_val("", V(src).shape, to, Lay::plain, V(src).batched);
        V(y).up32 = to == OSG_F32;
        const osg_dtype from = V(src).dtype;
        const long n = P.total_elems(src);
        P.add_step(std::string(to == OSG_F32 ? "upcast " : "downcast ") + V(src).name, {src}, {y}, [=, this] {
            be.check(be.api.osg_convert(be.ctx, from, to, P.ptr(src), P.ptr(y), n, 1.f, 0), "osg_convert");
        });
        cache[v] = y;
        return y;
    }
    int in_val(const Tensor& t) {
        const int v = P.ensure_dense(in_val_raw(t));
        if (cur_up && V(v).dtype == OSG_F16) return cast_val(v, OSG_F32);
        if (!cur_up && V(v).up32) return cast_val(v, OSG_F16);
        return v;
    }

    void check_out(const Operation& op, const Shape& got, size_t idx = 0) {
        // the reference's per-op self check (check_output_shape, :3070)
        const auto& want = op.m_output[idx].m_shape;
        bool ok = want.size() == got.size();
        for (size_t i = 0; ok && i < got.size(); i++) ok = (long)want[i] == got[i] || (m.m_support_dynamic_shapes && want[i] == 0);
        if (!ok && !(m.m_support_dynamic_shapes && want.empty()))
            throw std::invalid_argument(op.m_type + ": unexpected shape of output. (" + op.m_name + ": computed " + shape_str(got) + ")");
    }

    int out_val(const Operation& op, const Shape& shape, Lay lay, bool batched, osg_dtype dt = OSG_F16, size_t idx = 0) {
        check_out(op, shape, idx);
        return P.new_val(op.m_output[idx].m_name, shape, dt, lay, batched);
    }

    long B(int v) { return V(v).batched ? N : 1; }

    bool upcast_op(const Operation& op) const { return P.fp16 && !P.u8 && m.m_requires_upcast && m.m_requires_upcast(op.m_type, op.m_name); }
    void lower_all() {
        plan_linear_groups();
        if ((m.m_requires_upcast || P.sdp_attn) && !indexed) {   // (the op list no longer changes: one index serves the whole lowering)
            dead.assign(ops().size(), 0);                         // (run_fusions compacted the list: the flags of the old positions mean nothing now)
            index_graph();
            indexed = true;
        }
        static const bool per_type = std::getenv("OSG_PLAN_TIMING") && std::atoi(std::getenv("OSG_PLAN_TIMING")) >= 2;   // host microseconds of the lowering per op type
        std::map<std::string, std::pair<double, int>> type_us;
        for (size_t i = 0; i < ops().size(); i++) {
            const auto t_op = std::chrono::steady_clock::now();
            struct Acc {
                std::map<std::string, std::pair<double, int>>& m; const std::string& t; std::chrono::steady_clock::time_point t0; bool on;
                ~Acc() { if (on) { auto& e = m[t]; e.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); e.second++; } }
            } acc{type_us, ops()[i].m_type, t_op, per_type};
            if (i + 1 == ops().size() && per_type) {
                fprintf(stderr, "[plan] lowering by op type (us, ops):");
                for (auto& e : type_us) fprintf(stderr, " %s %.0f/%d", e.first.c_str(), e.second.first, e.second.second);
                fprintf(stderr, "\n");
            }
            if (group_of.count((int)i)) lower_group_member(ops()[i], (int)i);
            else if (ops()[i].m_type == "osg.RMSNorm") lower(ops()[i]);   // (fp32 inside by construction: fuse_rms_norm only fuses fully flagged chains)
            else if (upcast_op(ops()[i])) {
                const Operation& op = ops()[i];
                const std::string& t = op.m_type;
                if (!try_host_eval(op)) {
                    if (!(t == "Pow" || t == "ReduceMean" || t == "Add" || t == "Sub" || t == "Mul" || t == "Div" || t == "Sqrt" || t == "Neg" || t == "Sigmoid" || t == "Erf" ||
                          t == "Sin" || t == "Cos"))
                        throw std::invalid_argument("Model::run: m_requires_upcast is not implemented on the HIP backend for operation " + t + " (" + op.m_name + ").");
                    cur_up = true;
                    try { lower(op); } catch (...) { cur_up = false; throw; }
                    cur_up = false;
                    // push_tensor: the fp32 result is rounded to fp16 unless the next op of the queue is its only consumer
                    auto it = P.by_name.find(op.m_output[0].m_name);
                    if (it != P.by_name.end() && V(it->second).up32) {
                        bool keep = i + 1 < ops().size() && use_count(op.m_output[0].m_name) == 1;
                        if (keep) {
                            keep = false;
                            for (auto& ti : ops()[i + 1].m_input) keep |= ti.m_name == op.m_output[0].m_name;
                        }
                        if (!keep) {
                            const int h = cast_val(it->second, OSG_F16);
                            V(h).name = op.m_output[0].m_name;
                            P.by_name[op.m_output[0].m_name] = h;
                        }
                    }
                }
            } else lower(ops()[i]);
            if (P.calibrate)      // m_range_data_calibrate: every op output is measured right after the launch(es) that produce it
                for (auto& o : ops()[i].m_output) {
                    auto it = P.by_name.find(o.m_name);
                    if (it != P.by_name.end() && V(it->second).dtype == OSG_F16 && !V(it->second).host_only)
                        P.calib.push_back(Plan::Calib{(int)std::max<size_t>(P.steps.size(), 1) - 1, ops()[i].m_name, it->second});
                }
        }
    }

    void lower(const Operation& op) {
        const std::string& t = op.m_type;
        if (try_host_eval(op)) return;
        if (P.u8) return lower_u8(op);
        if (t == "Gather") return lower_gather(op);
        if (t == "Cast") return lower_cast(op);
        if (t == "Conv") {
            conv1d_out = Conv1dOut{};
            lower_conv(op);
            if (conv1d_out.y >= 0) {   // Conv1D: NHWC [1, Lo, 1, O] -> plain [1, O, Lo, 1] = the graph's [1, O, Lo]
                check_out(op, {1, conv1d_out.Cout, conv1d_out.Lo});
                P.alias(P.ensure_plain(conv1d_out.y), {1, conv1d_out.Cout, conv1d_out.Lo}, Lay::plain, op.m_output[0].m_name);
            }
            return;
        }
        if (t == "MatMul") return lower_matmul(op);
        if (t == "osg.Linear") return lower_linear(op);
        if (t == "Gemm") return lower_gemm(op);
        if (t == "Add" || t == "Sub" || t == "Mul" || t == "Div") return lower_binary(op);
        if (t == "Sigmoid" || t == "Erf" || t == "Sqrt" || t == "Sin" || t == "Cos" || t == "Neg" || t == "osg.SiLU") return lower_unary(op);
        if (t == "Pow") return lower_pow(op);
        if (t == "InstanceNormalization") return lower_instance_norm(op);
        if (t == "osg.GroupNorm") return lower_group_norm(op);
        if (t == "osg.LayerNorm") return lower_layer_norm(op);
        if (t == "osg.GEGLU") return lower_geglu(op);
        if (t == "osg.Attention") return lower_attention(op);
        if (t == "osg.TBlockTail") return lower_tblock_tail(op);
        if (t == "AttentionFusedOps") return lower_attention_fused_ops(op);
        if (t == "ScaledDotProductAttention") return lower_sdpa(op);
        if (t == "Expand") return lower_expand(op);
        if (t == "osg.RMSNorm") return lower_rms_norm(op);
        if (t == "osg.RoPE") return lower_rope(op);
        if (t == "ReduceMean") return lower_reduce_mean(op);
        if (t == "Softmax") return lower_softmax(op);
        if (t == "Reshape") return lower_reshape(op);
        if (t == "Flatten") return lower_flatten(op);
        if (t == "Unsqueeze" || t == "Squeeze") return lower_squeeze(op, t == "Unsqueeze");
        if (t == "Transpose") return lower_transpose(op);
        if (t == "Concat") return lower_concat(op);
        if (t == "Split") return lower_split(op);
        if (t == "Slice") return lower_slice(op);
        if (t == "Resize") return lower_resize(op);
        if (t == "MaxPool") return lower_maxpool(op);
        throw std::invalid_argument("Model::run: operation not implemented on the HIP backend: " + t);
    }

    static void need(const Operation& op, bool cond, const char* msg) {
        if (!cond) throw std::invalid_argument(op.m_type + ": " + msg);
    }

    // Conv (reference :4494-4707 -> XnnPack::convolution :1292): group 1, dilation 1, pads re-centred (:1315-1329)
    // GEMM-shaped steps ([M,K] x [N,K]^T, plain epilogue) that can ALSO emit the partial row statistics of their output when a folded
    // LayerNorm turns out to consume it (osg_gemm_rowstats): keyed by the root val they write
    struct RsProducer { size_t step; int a, w, bias, res, y; long M, N, K; };
    std::map<int, RsProducer> rs_producers;
    void note_rs_producer(int a, int w, int bias, int res, int y, long M, long Nn, long K) {
        if (P.fusion < 2 || !P.fuse_ln_gemm || Nn % 32 || K % 64 || Nn > 1280) return;
        rs_producers[P.root_of(y)] = RsProducer{P.steps.size() - 1, a, w, bias, res, y, M, Nn, K};
    }
    // switch the producer of x to the row-statistics variant; returns the [M, N/32, 2] fp32 val or -1
    int upgrade_rs_producer(int x, long rows, long C) {
        auto it = rs_producers.find(P.root_of(x));
        if (it == rs_producers.end()) return -1;
        const RsProducer r = it->second;
        if (r.M != rows || r.N != C || V(x).view_off != 0 || r.step >= P.steps.size()) return -1;
        if (viewed_steps.count(r.step)) return -1;   // (its output also goes into a Concat slot: the row-statistics launch has no output views)
        {
            auto cpit = conv_producers.find(P.root_of(r.y));
            if (cpit != conv_producers.end() && (cpit->second.out->sink[0].off >= 0 || cpit->second.out->sink[1].off >= 0)) return -1;   // (a GroupNorm reads its statistics from this launch's epilogue)
        }
        if (V(r.y).batched != V(x).batched) return -1;
        int rs = P.new_val("", {r.M / (V(x).batched ? N : 1), C / 32, 2}, OSG_F32, Lay::plain, V(x).batched);
        Step& st = P.steps[r.step];
        st.writes.push_back(rs);
        st.what += " +rowstats";
        const int a = r.a, w = r.w, bias = r.bias, res = r.res, y = r.y;
        const long M = r.M, Nn = r.N, K = r.K;
        const std::string what = st.what;
        st.run = [=, this] {
            be.check(be.api.osg_gemm_rowstats(be.ctx, P.ptr(a), P.ptr(w), bias >= 0 ? P.ptr(bias) : nullptr, bias >= 0 ? P.vals[bias].dtype : OSG_F16,
                                              res >= 0 ? P.ptr(res) : nullptr, P.ptr(y), (int)M, (int)Nn, (int)K, OSG_ACT_NONE, (float*)P.ptr(rs)),
                     what.c_str());
        };
        rs_producers.erase(it);
        conv_producers.erase(P.root_of(y));   // (the launch is the row-statistics GEMM now: no output views)
        return rs;
    }

    // ---- output views of the plain f16 convolutions (round 3): a Concat of NHWC tensors along the channels whose operands come straight out of
    // convolutions is not launched at all -- each producer stores its result into ITS column slice of the concatenated buffer (osg_conv2d_nhwc_v):
    // as its only destination when the Concat is its only reader, next to the dense tensor when other layers read it too (the skip connections
    // of the UNet: 12 copy launches and 2 x the tensors' bytes per pass).  The launch closure reads its destinations from `ConvOut` at run time.
    // sink[k]: GroupNorm statistics of what the launch stores to dst (k = 0) / dst2 (k = 1), added up by its epilogue into the plan's statistics block at
    // `off` (osg_set_stat_sinks; lower_group_norm arms the producers of the tensor it normalises)
    struct StatSinkRef { long off = -1; int groups = 0, cpg = 0, ch_off = 0; };
    struct ConvOut { int dst; long dst_ld = 0; size_t dst_off = 0; int dst2 = -1; long dst2_ld = 0; size_t dst2_off = 0; StatSinkRef sink[2]; int sink_hw = 0; bool no_sinks = false; };   // (no_sinks: a producer whose epilogue cannot add statistics up, osg.TBlockTail)
    struct ConcatPart { std::shared_ptr<ConvOut> out; int slot; long ch_off; size_t step; };
    std::map<int, std::vector<ConcatPart>> concat_parts;   // root val of a Concat output whose operands ALL go there by output views -> the convolutions that fill it
    struct ConvProducer { size_t step; std::shared_ptr<ConvOut> out; int y; };
    std::map<int, ConvProducer> conv_producers;   // root val of a convolution's output -> its step
    std::set<size_t> viewed_steps;                // steps whose destinations were redirected (their launch must stay the view-aware one)

    struct Conv1dOut { int y = -1; long Cout = 0, Lo = 0; } conv1d_out;
    void lower_conv(const Operation& op) {
        const bool has_res = attr(op, "osg_residual") != nullptr;
        const bool has_ib = attr(op, "osg_image_bias") != nullptr;
        const osg_act cact = attr(op, "osg_act") ? OSG_ACT_SILU : OSG_ACT_NONE;
        const size_t nin = op.m_input.size();
        need(op, nin == 2 || nin == 3 || (has_res && nin == 4) || (has_ib && nin == 5), "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        std::vector<int> dil = {1, 1}, ks, pads = {0, 0, 0, 0}, strides = {1, 1};
        int group = 1;
        for (auto& a : op.m_attributes) {
            if (a.first == "dilations") dil = int_list(a.second);
            else if (a.first == "group") group = std::stoi(a.second);
            else if (a.first == "kernel_shape") ks = int_list(a.second);
            else if (a.first == "pads") pads = int_list(a.second);
            else if (a.first == "strides") strides = int_list(a.second);
            else if (a.first == "osg_residual" || a.first == "osg_image_bias" || a.first == "osg_act") {}
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        int x = in_val(op.m_input[0]);
        // Conv1D (reference :4521-4544): run as the 2-D convolution over [N, C, L, 1] -- dilations / kernel_shape gain a 1, pads [b, e] become [b, 0, e, 0], the
        // stride is repeated -- and the [N, O, Lo, 1] result is handed on as [N, O, Lo]
        const bool is1d = V(x).shape.size() == 3;
        // (the reference keys the lift on dilations.size() == 1, :4521; here the input's rank decides and a dilations attribute that disagrees with it is refused)
        for (auto& a : op.m_attributes)
            if (a.first == "dilations") need(op, (dil.size() == 1) == is1d, "invalid dilations attribute value.");
        if (is1d) {
            need(op, !has_res && !has_ib, "Conv1D with fused epilogue inputs not implemented on the HIP backend.");
            bool have_pads = false, have_strides = false, have_dil = false;
            for (auto& a : op.m_attributes) { have_pads |= a.first == "pads"; have_strides |= a.first == "strides"; have_dil |= a.first == "dilations"; }
            if (have_dil) { need(op, dil.size() == 1, "invalid dilations attribute value."); dil.push_back(1); }
            if (!ks.empty()) { need(op, ks.size() == 1, "invalid kernel_shape attribute value."); ks.push_back(1); }
            if (have_pads) { need(op, pads.size() == 2, "invalid pads attribute value."); pads = {pads[0], 0, pads[1], 0}; }
            if (have_strides) { need(op, strides.size() == 1, "invalid strides attribute value."); strides.push_back(strides[0]); }
            Shape s4 = V(x).shape;
            s4.push_back(1);
            x = P.alias(P.ensure_plain(x), s4, Lay::plain);
        }
        need(op, V(x).shape.size() == 4, "invalid shape of input.");
        for (int d : dil) need(op, d == 1, "dilations != 1 not supported (not implemented).");
        need(op, group == 1, "group != 1 not supported (not implemented).");
        need(op, pads.size() == 4 && strides.size() == 2, "invalid pads/strides.");
        x = P.ensure_nhwc(x);
        int w = in_val(op.m_input[1]);
        const Shape ws = V(w).shape;  // [O, I, kh, kw] as named in model.txt; data is OHWI
        need(op, V(w).is_const && V(w).lay == Lay::nhwc && ws.size() == 4, "weights must be a static *_nchw.bin tensor.");
        const Shape xs = V(x).shape;
        need(op, xs[0] == 1, "batch size must be 1 (push several samples instead).");
        const long Cin = xs[1], H = xs[2], W = xs[3], Cout = ws[0], KH = ws[2], KW = ws[3];
        need(op, ws[1] == Cin, "invalid shape of weights.");
        if (!ks.empty()) need(op, ks.size() == 2 && ks[0] == KH && ks[1] == KW, "kernel_shape does not match the weights.");
        const int ph = pads[0] + pads[2], pw = pads[1] + pads[3];
        const int pt = ph / 2, pb = ph - pt, pl = pw / 2, pr = pw - pl;
        const long Ho = (H + ph - KH) / strides[0] + 1, Wo = (W + pw - KW) / strides[1] + 1;
        int bias = -1;
        if (nin >= 3 && !op.m_input[2].m_name.empty()) {
            bias = in_val(op.m_input[2]);
            need(op, V(bias).numel() == Cout, "invalid shape of bias.");
        }
        int res = -1;
        if (has_res) res = P.ensure_nhwc(in_val(op.m_input[3]));
        int ib = -1;      // per-image channel bias [1,Cout] (may be a column view of the merged time-embedding projection)
        long ib_ld = 0;
        if (has_ib) {
            ib = in_val_raw(op.m_input[4]);
            need(op, V(ib).numel() == Cout && V(ib).dtype == OSG_F16 && V(ib).batched == V(x).batched, "invalid image bias.");
            ib_ld = V(ib).ld ? V(ib).ld : Cout;
        }
        int y = is1d ? P.new_val("", {1, Cout, Ho, Wo}, OSG_F16, Lay::nhwc, V(x).batched) : out_val(op, {1, Cout, Ho, Wo}, Lay::nhwc, V(x).batched);
        if (is1d) conv1d_out = Conv1dOut{y, Cout, Ho};   // (lower() hands the [1, O, Lo, 1] result on as the graph's [1, O, Lo] once the launch is planned)
        const long nb = B(x);
        const int sh = strides[0], sw = strides[1];
        std::vector<int> reads = {x, w};
        if (bias >= 0) reads.push_back(bias);
        if (res >= 0) reads.push_back(res);
        if (ib >= 0) reads.push_back(ib);
        if (V(w).dtype == OSG_U8) {
            const float qs = V(w).qscale;
            const int qz = V(w).qzp;
            P.add_step("Conv w8 " + op.m_name, reads, {y}, [=, this] {
                be.check(be.api.osg_conv2d_nhwc_w8(be.ctx, P.ptr(x), P.ptr(w), qs, qz, bias >= 0 ? P.ptr(bias) : nullptr,
                                                   bias >= 0 ? P.vals[bias].dtype : OSG_F16, ib >= 0 ? P.ptr(ib) : nullptr, ib_ld,
                                                   res >= 0 ? P.ptr(res) : nullptr, P.ptr(y), (int)nb, (int)H, (int)W, (int)Cin, (int)Cout,
                                                   (int)KH, (int)KW, sh, sw, pt, pl, pb, pr, cact),
                         "Conv");
            });
            P.steps.back().flops = 2.0 * nb * Ho * Wo * Cout * KH * KW * Cin;
            return;
        }
        auto co = std::make_shared<ConvOut>();
        co->dst = y;
        P.add_step("Conv " + op.m_name, reads, {y}, [=, this] {
            const ConvOut& o = *co;
            if (o.sink[0].off >= 0 || o.sink[1].off >= 0)
                be.check(be.api.osg_set_stat_sinks(be.ctx, o.sink[0].off >= 0 ? P.gn_stats + o.sink[0].off : nullptr, o.sink[0].groups, o.sink[0].cpg, o.sink[0].ch_off,
                                                   o.sink[1].off >= 0 ? P.gn_stats + o.sink[1].off : nullptr, o.sink[1].groups, o.sink[1].cpg, o.sink[1].ch_off, o.sink_hw),
                         "osg_set_stat_sinks");
            be.check(be.api.osg_conv2d_nhwc_v(be.ctx, OSG_F16, P.ptr(x), P.ptr(w), bias >= 0 ? P.ptr(bias) : nullptr,
                                              bias >= 0 ? P.vals[bias].dtype : OSG_F16, ib >= 0 ? P.ptr(ib) : nullptr, ib_ld,
                                              res >= 0 ? P.ptr(res) : nullptr, (char*)P.ptr(o.dst) + o.dst_off, o.dst_ld,
                                              o.dst2 >= 0 ? (char*)P.ptr(o.dst2) + o.dst2_off : nullptr, o.dst2_ld, (int)nb, (int)H, (int)W, (int)Cin, (int)Cout,
                                              (int)KH, (int)KW, sh, sw, pt, pl, pb, pr, cact),
                     "Conv");
        });
        P.steps.back().flops = 2.0 * nb * Ho * Wo * Cout * KH * KW * Cin;
        if (V(w).dtype == OSG_F16) conv_producers[P.root_of(y)] = ConvProducer{P.steps.size() - 1, co, y};
        if (KH == 1 && KW == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && ib < 0 && cact == OSG_ACT_NONE)   // a 1x1 convolution IS a GEMM over the pixels (OHWI == [N,K])
            note_rs_producer(x, w, bias, res, y, nb * Ho * Wo, Cout, Cin);
    }

    // resident [K,N] weight -> [N,K] (done once, at plan time)
    int weight_nk(int w) {
        Val& wv = V(w);
        if (wv.as_nhwc >= 0) return wv.as_nhwc;  // reuse the slot: "K-contiguous twin"
        const long K = wv.shape[0], Nn = wv.shape[1];
        int t = P.new_val("", {Nn, K}, OSG_F16, Lay::plain, false);
        Val& tv = V(t);
        tv.is_const = true;
        tv.name = V(w).name + "|nk";
        size_t bytes = (size_t)K * Nn * 2;
        bool fresh;
        tv.dptr = P.const_alloc(tv.name, bytes, &fresh);
        if (fresh) be.check(be.api.osg_transpose_kn_to_nk(be.ctx, OSG_F16, P.ptr(w), tv.dptr, (int)K, (int)Nn), "osg_transpose_kn_to_nk");
        V(w).as_nhwc = t;
        return t;
    }

    // [K,N] uint8 codes -> [N,K] (done once, at plan time)
    int weight_nk_u8(int w) {
        if (V(w).as_nk_u8 >= 0) return V(w).as_nk_u8;
        const long K = V(w).shape[0], Nn = V(w).shape[1];
        int t = P.new_val("", {Nn, K}, OSG_U8, Lay::plain, false);
        V(t).is_const = true;
        V(t).name = V(w).name + "|nk_u8";
        V(t).qscale = V(w).qscale;
        V(t).qzp = V(w).qzp;
        bool fresh;
        V(t).dptr = P.const_alloc(V(t).name, (size_t)K * Nn, &fresh);
        if (fresh) {
            long shape[2] = {K, Nn};
            int perm[2] = {1, 0};
            be.check(be.api.osg_transpose(be.ctx, 1, P.ptr(w), V(t).dptr, 2, shape, perm), "osg_transpose");
            be.check(be.api.osg_sync(be.ctx), "osg_sync");
        }
        V(w).as_nk_u8 = t;
        return t;
    }

    void emit_gemm_w8(const std::string& what, int a, int w, int bias, int res, int y, long M, long Nn, long K) {
        const int wq = weight_nk_u8(w);
        const float qs = V(wq).qscale;
        const int qz = V(wq).qzp;
        std::vector<int> reads = {a, wq};
        if (bias >= 0) reads.push_back(bias);
        if (res >= 0) reads.push_back(res);
        P.add_step(what, reads, {y}, [=, this] {
            be.check(be.api.osg_gemm_w8(be.ctx, P.ptr(a), P.ptr(wq), qs, qz, bias >= 0 ? P.ptr(bias) : nullptr,
                                        bias >= 0 ? P.vals[bias].dtype : OSG_F16, res >= 0 ? P.ptr(res) : nullptr, P.ptr(y), (int)M, (int)Nn, (int)K,
                                        OSG_ACT_NONE),
                     what.c_str());
        });
        P.steps.back().flops = 2.0 * M * Nn * K;
    }

    struct LnFold { int x, g, b; float eps; long C; int rs; };   // rs: partial row statistics handed over by the producer of x (-1: none)

    void emit_gemm(const std::string& what, int a, int wnk, int bias, int res, int y, long M, long Nn, long K, long batch, long sa, long sb,
                   long sc, int b_is_nk, osg_act act_ = OSG_ACT_NONE) {
        std::vector<int> reads = {a, wnk};
        if (bias >= 0) reads.push_back(bias);
        if (res >= 0) reads.push_back(res);
        P.add_step(what, reads, {y}, [=, this] {
            be.check(be.api.osg_gemm(be.ctx, OSG_F16, P.ptr(a), P.ptr(wnk), b_is_nk, bias >= 0 ? P.ptr(bias) : nullptr,
                                     bias >= 0 ? P.vals[bias].dtype : OSG_F16, res >= 0 ? P.ptr(res) : nullptr, P.ptr(y), (int)M, (int)Nn,
                                     (int)K, (int)batch, sa, sb, sc, act_),
                     what.c_str());
        });
        P.steps.back().flops = 2.0 * M * Nn * K * batch;
        if (b_is_nk && batch == 1 && V(y).ld == 0 && act_ == OSG_ACT_NONE) note_rs_producer(a, wnk, bias, res, y, M, Nn, K);
    }

    // ---- merged projections: osg.Linear / Gemm ops that read the SAME activation and whose results are only consumed through
    // strided views (attention Q/K/V operands, conv image bias) run as ONE GEMM over the concatenated [sum N_i, K] weight:
    // self-attention Q|K|V (3 launches -> 1), every cross-attention K|V of the net (they all read the text context: 32 -> 1),
    // the 22 time-embedding projections of the resnet blocks (22 -> 1).
    struct LinGroup {
        std::vector<int> members;   // op indices, file order
        std::vector<long> off;      // column offset of each member inside the merged output
        long ntot = 0;
        int y = -1;                 // merged output val [rows, ntot]
    };
    bool indexed = false;   // index_graph() describes the op list being lowered
    std::vector<LinGroup> groups;
    std::map<int, std::pair<int, int>> group_of;   // op index -> (group, slot)

    void plan_linear_groups() {
        if (P.fusion < 2 || P.stream_weights) return;   // streamed weights are consumed as the provider hands them over: no merged copies
        dead.assign(ops().size(), 0);
        index_graph();
        indexed = true;
        std::map<std::string, std::vector<int>> by_key;
        for (size_t i = 0; i < ops().size(); i++) {
            const Operation& op = ops()[i];
            const bool lin = op.m_type == "osg.Linear", gemm = op.m_type == "Gemm";
            if (!lin && !gemm) continue;
            if (attr(op, "osg_residual") || op.m_output.size() != 1 || !act(op.m_input[0])) continue;
            const Val* w = cval(op.m_input[1]);
            if (!w || w->shape.size() != 2 || w->dtype != OSG_F16) continue;
            const bool has_bias = op.m_input.size() > 2 && !op.m_input[2].m_name.empty();
            if (gemm && (!has_bias || !op.m_attributes.empty())) continue;
            if (has_bias && cval(op.m_input[2]) && cval(op.m_input[2])->dtype != OSG_F16) continue;
            // every consumer must understand a leading dimension
            bool ok = use_count(op.m_output[0].m_name) < 1000;
            auto cit = consumers.find(op.m_output[0].m_name);
            if (cit == consumers.end() || cit->second.empty()) ok = false;
            else
                for (int c : cit->second) {
                    const Operation& co = ops()[c];
                    if (co.m_type == "osg.Attention" || co.m_type == "osg.TBlockTail" || co.m_type == "osg.QAttention") continue;   // (all read k / v through (pointer, row pitch))
                    if (co.m_type == "Conv" && attr(co, "osg_image_bias") && co.m_input.size() >= 5 && co.m_input[4].m_name == op.m_output[0].m_name) continue;
                    ok = false;
                }
            if (!ok) continue;
            by_key[op.m_type + "|" + op.m_input[0].m_name + "|" + std::to_string(w->shape[0]) + (has_bias ? "|b" : "|-")].push_back((int)i);
        }
        for (auto& kv : by_key) {
            if (kv.second.size() < 2) continue;
            LinGroup g;
            for (int i : kv.second) {
                g.members.push_back(i);
                g.off.push_back(g.ntot);
                g.ntot += cval(ops()[i].m_input[1])->shape[1];
            }
            for (size_t s2 = 0; s2 < g.members.size(); s2++) group_of[g.members[s2]] = {(int)groups.size(), (int)s2};
            groups.push_back(std::move(g));
        }
    }

    // emits the merged GEMM the first time one of its members is lowered; returns the member's column view
    int lower_group_member(const Operation& op, int op_index) {
        auto [gi, slot] = group_of.at(op_index);
        LinGroup& g = groups[gi];
        const Val* w0 = cval(op.m_input[1]);
        const long K = w0->shape[0];
        if (g.y < 0) {
            auto lnit = ln_deferred.find(op.m_input[0].m_name);
            const LnFold* lnf = lnit == ln_deferred.end() ? nullptr : &lnit->second;
            int a = lnf ? lnf->x : P.ensure_plain(in_val(op.m_input[0]));
            const Shape as = V(a).shape;
            need(op, !as.empty() && as.back() == K, "invalid shape of inputs.");
            // concatenated [ntot, K] weight and [ntot] bias, built once from the resident per-op tensors
            int wcat = P.new_val("", {g.ntot, K}, OSG_F16, Lay::plain, false);
            V(wcat).is_const = true;
            V(wcat).name = "merged|" + op.m_input[0].m_name + "|" + ops()[g.members[0]].m_name + "|" + std::to_string(g.members.size()) + (lnf ? "|ln" : "");   // (a folded copy has a tag of its own)
            bool fresh_w, fresh_b = false;
            V(wcat).dptr = P.const_alloc(V(wcat).name, (size_t)g.ntot * K * 2, &fresh_w);
            const bool has_bias = op.m_input.size() > 2 && !op.m_input[2].m_name.empty();
            int bcat = -1;
            if (has_bias) {
                bcat = P.new_val("", {g.ntot}, OSG_F16, Lay::plain, false);
                V(bcat).is_const = true;
                V(bcat).dptr = P.const_alloc(V(wcat).name + "|bias", (size_t)g.ntot * 2, &fresh_b);
            }
            for (size_t s2 = 0; (fresh_w || fresh_b) && s2 < g.members.size(); s2++) {
                const Operation& mo = ops()[g.members[s2]];
                int wnk = weight_nk(in_val(mo.m_input[1]));
                const long Ni = V(wnk).shape[0];
                be.check(be.api.osg_copy(be.ctx, (char*)V(wcat).dptr + (size_t)g.off[s2] * K * 2, P.ptr(wnk), (size_t)Ni * K * 2), "osg_copy");
                if (has_bias) {
                    int b = in_val(mo.m_input[2]);
                    need(mo, V(b).numel() == Ni, "invalid shape of bias.");
                    be.check(be.api.osg_copy(be.ctx, (char*)V(bcat).dptr + (size_t)g.off[s2] * 2, P.ptr(b), (size_t)Ni * 2), "osg_copy");
                }
            }
            be.check(be.api.osg_sync(be.ctx), "osg_sync");
            Shape ys = as;
            ys.back() = g.ntot;
            g.y = P.new_val("", ys, OSG_F16, Lay::plain, V(a).batched);
            const long M = prod(as) / K * B(a);
            if (lnf) {
                auto [c1, c2] = ln_fold_weight(*lnf, wcat, bcat, P.ptr(wcat), V(wcat).name);   // the concatenated copy is private: fold in place
                emit_gemm_ln("Linear ln+ merged(" + std::to_string(g.members.size()) + ") " + op.m_name, *lnf, wcat, c1, c2, -1, g.y, M, g.ntot, K, OSG_ACT_NONE);
            } else
            emit_gemm("Linear merged(" + std::to_string(g.members.size()) + ") " + op.m_name, a, wcat, bcat, -1, g.y, M, g.ntot, K, 1, 0, 0, 0, 1);
        }
        Shape os = V(g.y).shape;
        os.back() = w0->shape[1];
        check_out(op, os);
        int v = P.alias(g.y, os, Lay::plain, op.m_output[0].m_name);
        V(v).ld = g.ntot;
        V(v).view_off = (size_t)g.off[slot] * 2;
        V(v).is_const = false;
        return v;
    }

    // MatMul with a static 2-D weight, optional fused bias / residual
    // [N,K] weight (and bias) re-ordered so that rows 32k..32k+15 are value columns 16k.. and rows 32k+16..32k+31 the matching gate
    // columns (N = 2C): what the GEGLU GEMM epilogue expects
    std::pair<int, int> geglu_interleave(int wnk, int bias, const std::string& suffix) {
        const long Nn = V(wnk).shape[0], K = V(wnk).shape[1], C = Nn / 2;
        int wi = P.new_val("", {Nn, K}, OSG_F16, Lay::plain, false);
        V(wi).is_const = true;
        V(wi).name = V(wnk).name + "|geglu" + suffix;
        bool fresh;
        V(wi).dptr = P.const_alloc(V(wi).name, (size_t)Nn * K * 2, &fresh);
        if (fresh) {
            be.check(be.api.osg_copy_2d(be.ctx, 2, P.ptr(wnk), 16 * K, 0, V(wi).dptr, 32 * K, 0, C / 16, 16 * K), "osg_copy_2d");
            be.check(be.api.osg_copy_2d(be.ctx, 2, P.ptr(wnk), 16 * K, C * K, V(wi).dptr, 32 * K, 16 * K, C / 16, 16 * K), "osg_copy_2d");
        }
        int bi = -1;
        if (bias >= 0) {
            bi = P.new_val("", {Nn}, OSG_F16, Lay::plain, false);
            V(bi).is_const = true;
            V(bi).dptr = P.const_alloc(V(wi).name + "|bias", (size_t)Nn * 2, &fresh);
            if (fresh) {
                be.check(be.api.osg_copy_2d(be.ctx, 2, P.ptr(bias), 16, 0, V(bi).dptr, 32, 0, C / 16, 16), "osg_copy_2d");
                be.check(be.api.osg_copy_2d(be.ctx, 2, P.ptr(bias), 16, C, V(bi).dptr, 32, 16, C / 16, 16), "osg_copy_2d");
            }
        }
        be.check(be.api.osg_sync(be.ctx), "osg_sync");
        return {wi, bi};
    }

    void lower_linear(const Operation& op) {
        auto lnit = ln_deferred.find(op.m_input[0].m_name);
        const LnFold* lnf = lnit == ln_deferred.end() ? nullptr : &lnit->second;
        if (attr(op, "osg_geglu")) {
            int a = lnf ? lnf->x : P.ensure_plain(in_val(op.m_input[0]));
            int w = in_val(op.m_input[1]);
            const Shape as = V(a).shape;
            const long K = V(w).shape[0], Nn = V(w).shape[1];
            need(op, !as.empty() && as.back() == K, "invalid shape of inputs.");
            int bias = op.m_input.size() > 2 && !op.m_input[2].m_name.empty() ? in_val(op.m_input[2]) : -1;
            auto [wi, bi] = geglu_interleave(weight_nk(w), bias, lnf ? ln_tag(*lnf, bias) : std::string());
            Shape os = as;
            os.back() = Nn / 2;
            int y = out_val(op, os, Lay::plain, V(a).batched);
            const long M = prod(as) / K * B(a);
            if (lnf) {
                auto [c1, c2] = ln_fold_weight(*lnf, wi, bi, P.ptr(wi), V(wi).name);   // the interleaved copy is private: fold in place
                emit_gemm_ln("Linear+GEGLU ln+ " + op.m_name, *lnf, wi, c1, c2, -1, y, M, Nn, K, OSG_ACT_GEGLU);
                return;
            }
            std::vector<int> reads = {a, wi};
            if (bi >= 0) reads.push_back(bi);
            const std::string what = "Linear+GEGLU " + op.m_name;
            P.add_step(what, reads, {y}, [=, this] {
                be.check(be.api.osg_gemm(be.ctx, OSG_F16, P.ptr(a), P.ptr(wi), 1, bi >= 0 ? P.ptr(bi) : nullptr, OSG_F16, nullptr, P.ptr(y), (int)M,
                                         (int)Nn, (int)K, 1, 0, 0, 0, OSG_ACT_GEGLU),
                         what.c_str());
            });
            P.steps.back().flops = 2.0 * M * Nn * K;
            return;
        }
        const bool has_res = attr(op, "osg_residual") != nullptr;
        int a = lnf ? lnf->x : P.ensure_plain(in_val(op.m_input[0]));
        int w = in_val(op.m_input[1]);
        const Shape as = V(a).shape;
        const long K = V(w).shape[0], Nn = V(w).shape[1];
        need(op, !as.empty() && as.back() == K, "invalid shape of inputs.");
        int bias = op.m_input.size() > 2 && !op.m_input[2].m_name.empty() ? in_val(op.m_input[2]) : -1;
        int res = has_res ? P.ensure_plain(in_val(op.m_input[3])) : -1;
        Shape os = as;
        os.back() = Nn;
        int y = out_val(op, os, Lay::plain, V(a).batched);
        const long M = prod(as) / K * B(a);
        if (lnf) {
            const int wnk = weight_nk(w);
            const int wf = private_copy(wnk, ln_tag(*lnf, bias));
            auto [c1, c2] = ln_fold_weight(*lnf, wnk, bias, P.ptr(wf), V(wf).name);
            emit_gemm_ln("Linear ln+ " + op.m_name, *lnf, wf, c1, c2, res, y, M, Nn, K, OSG_ACT_NONE);
            return;
        }
        if (V(w).dtype == OSG_U8) emit_gemm_w8("Linear w8 " + op.m_name, a, w, bias, res, y, M, Nn, K);
        else if (P.stream_weights) emit_gemm("Linear " + op.m_name, a, w, bias, res, y, M, Nn, K, 1, 0, 0, 0, 0);   // [K,N] as streamed; the kernel re-lays it out
        else emit_gemm("Linear " + op.m_name, a, weight_nk(w), bias, res, y, M, Nn, K, 1, 0, 0, 0, 1);
    }

    // MatMul (reference :5669-5861): [.., M,K] x [K,N] (static weight, broadcast) or batched [n,M,K] x [n,K,N]
    void lower_matmul(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        int b = in_val(op.m_input[1]);
        if (V(b).is_const && V(b).shape.size() == 2) return lower_linear(op);
        int a = P.ensure_plain(in_val(op.m_input[0]));
        b = P.ensure_plain(b);
        Shape as = V(a).shape, bs = V(b).shape;
        bool lead1 = false;
        if (as.size() == 4 && as[0] == 1) { as.erase(as.begin()); lead1 = true; }
        if (bs.size() == 4 && bs[0] == 1) bs.erase(bs.begin());
        need(op, as.size() == 3 && bs.size() == 3 && as[0] == bs[0] && as[2] == bs[1], "invalid shape of inputs.");
        const long n = as[0], M = as[1], K = as[2], Nn = bs[2];
        Shape os = {n, M, Nn};
        if (lead1) os.insert(os.begin(), 1);
        const bool batched = V(a).batched || V(b).batched;
        int y = out_val(op, os, Lay::plain, batched);
        const long batch = n * (batched ? N : 1);
        // an operand that is not batched is re-read for every sample: stride 0 across samples is not expressible with one
        // stride, so require both or neither when N > 1
        need(op, N == 1 || (V(a).batched == V(b).batched), "mixing per-sample and shared dynamic operands is not implemented.");
        emit_gemm("MatMul " + op.m_name, a, b, -1, -1, y, M, Nn, K, batch, M * K, K * Nn, M * Nn, 0);
    }

    // Gemm (reference :4300-4375): A[1,K] x B[K,N] + bias, no alpha/beta/trans
    void lower_gemm(const Operation& op) {
        need(op, op.m_input.size() == 3, "wrong number of inputs. 2 inputs case not implemented.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        for (auto& a : op.m_attributes) {
            if (a.first == "alpha" || a.first == "beta") need(op, std::stof(a.second) == 1.0f, (a.first + " != 1 case not implemented.").c_str());
            else if (a.first == "transA" || a.first == "transB") need(op, std::stoi(a.second) == 0, (a.first + " != 0 case not implemented.").c_str());
            else if (a.first == "osg_act") {}
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        int a = P.ensure_plain(in_val(op.m_input[0]));
        int w = in_val(op.m_input[1]);
        int bias = in_val(op.m_input[2]);
        need(op, V(w).is_const && V(w).shape.size() == 2 && V(a).shape.size() == 2, "not implemented (shape of inputs).");
        const long M = V(a).shape[0], K = V(a).shape[1], Nn = V(w).shape[1];
        need(op, V(w).shape[0] == K, "invalid shape of inputs.");
        need(op, M == 1 && V(bias).numel() == Nn, "invalid shape of bias.");
        int y = out_val(op, {M, Nn}, Lay::plain, V(a).batched);
        const osg_act gact = attr(op, "osg_act") ? OSG_ACT_SILU : OSG_ACT_NONE;   // (fuse_gemm_act only marks f16-weight Gemms)
        if (V(w).dtype == OSG_U8) emit_gemm_w8("Gemm w8 " + op.m_name, a, w, bias, -1, y, M * B(a), Nn, K);
        else if (P.stream_weights) emit_gemm("Gemm " + op.m_name, a, w, bias, -1, y, M * B(a), Nn, K, 1, 0, 0, 0, 0, gact);
        else emit_gemm("Gemm " + op.m_name, a, weight_nk(w), bias, -1, y, M * B(a), Nn, K, 1, 0, 0, 0, 1, gact);
    }

    // Add/Sub/Mul/Div with NumPy broadcasting (reference :3906-4000, :5056-5175, :5394-5477, :5605-5668)
    void lower_binary(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        int a = in_val(op.m_input[0]), b = in_val(op.m_input[1]);
        const osg_dtype adt = cur_up ? OSG_F32 : OSG_F16;   // (a flagged op runs in fp32 on upcast operands, m_requires_upcast)
        need(op, V(a).dtype == adt && V(b).dtype == adt, "wrong data type of inputs (only the arithmetic type is supported on the device).");
        const osg_binary_kind kind = op.m_type == "Add" ? OSG_BIN_ADD : op.m_type == "Sub" ? OSG_BIN_SUB : op.m_type == "Mul" ? OSG_BIN_MUL : OSG_BIN_DIV;
        // logical output shape (right-aligned broadcast, :855-876)
        const Shape as = V(a).shape, bs = V(b).shape;
        const size_t rank = std::max(as.size(), bs.size());
        Shape os(rank);
        for (size_t i = 0; i < rank; i++) {
            long da = i + as.size() >= rank ? as[i + as.size() - rank] : 1;
            long db = i + bs.size() >= rank ? bs[i + bs.size() - rank] : 1;
            need(op, da == db || da == 1 || db == 1, "shapes are not broadcastable.");
            os[i] = std::max(da, db);
        }
        const bool batched = V(a).batched || V(b).batched;
        // ---- NHWC-aware fast paths: keep the conv layout when the other operand is per-channel / same-layout ----
        auto per_channel = [&](int v, long C) {  // logical [C,1,1] / [1,C,1,1] / scalar
            const Shape& s = V(v).shape;
            if (V(v).lay == Lay::nhwc) return false;
            if (V(v).numel() == 1) return true;
            if (V(v).numel() != C) return false;
            return (s.size() == 3 && s[0] == C) || (s.size() == 4 && s[1] == C);
        };
        Lay olay = Lay::plain;
        Shape pa, pb;  // physical shapes handed to the kernel (without the batch dim)
        if (V(a).lay == Lay::nhwc || V(b).lay == Lay::nhwc) {
            int x = V(a).lay == Lay::nhwc ? a : b, o = x == a ? b : a;
            const Shape xs = V(x).shape;
            const long C = xs[1], HW = xs[2] * xs[3];
            if (V(o).lay == Lay::nhwc && V(o).shape == xs) {
                olay = Lay::nhwc;
                pa = pb = {HW, C};
            } else if (per_channel(o, C) && os == xs) {
                olay = Lay::nhwc;
                Shape px = {HW, C}, po = {1, V(o).numel() == 1 ? 1 : C};
                pa = x == a ? px : po;
                pb = x == a ? po : px;
            } else {
                a = P.ensure_plain(a);
                b = P.ensure_plain(b);
            }
        }
        if (olay == Lay::plain) { pa = V(a).shape; pb = V(b).shape; }
        int y = out_val(op, os, olay, batched, adt);
        V(y).up32 = adt == OSG_F32;
        // prepend the sample dim
        const size_t prank = std::max(pa.size(), pb.size()) + 1;
        need(op, prank <= 6, "rank too large for the device broadcast kernel.");
        std::vector<long> sa(prank, 1), sb(prank, 1);
        for (size_t i = 0; i < pa.size(); i++) sa[prank - pa.size() + i] = pa[i];
        for (size_t i = 0; i < pb.size(); i++) sb[prank - pb.size() + i] = pb[i];
        sa[0] = B(a);
        sb[0] = B(b);
        P.add_step(op.m_type + " " + op.m_name, {a, b}, {y}, [=, this] {
            be.check(be.api.osg_binary(be.ctx, adt, kind, P.ptr(a), sa.data(), P.ptr(b), sb.data(), P.ptr(y), (int)prank), op.m_type.c_str());
        });
    }

    void lower_rms_norm(const Operation& op) {
        const int x = P.ensure_plain(in_val(op.m_input[0])), w = in_val(op.m_input[1]);
        need(op, V(x).dtype == OSG_F16 && V(w).dtype == OSG_F16, "wrong data type of input.");
        const Shape s = V(x).shape;
        need(op, !s.empty() && V(w).numel() == s.back(), "invalid shape of the weight.");
        const float eps = std::stof(*attr(op, "epsilon"));
        const int y = out_val(op, s, Lay::plain, V(x).batched);
        const long C = s.back(), rows = P.total_elems(x) / C;
        P.add_step("RMSNorm " + op.m_name, {x, w}, {y}, [=, this] { be.check(be.api.osg_rms_norm(be.ctx, OSG_F16, P.ptr(x), P.ptr(w), P.ptr(y), rows, (int)C, eps), "RMSNorm"); });
    }

    void lower_rope(const Operation& op) {
        const int x = P.ensure_plain(in_val(op.m_input[0])), cs = P.ensure_plain(in_val(op.m_input[1])), sn = P.ensure_plain(in_val(op.m_input[2]));
        need(op, V(x).dtype == OSG_F16 && V(cs).dtype == OSG_F16 && V(sn).dtype == OSG_F16, "wrong data type of input.");
        const Shape s = V(x).shape;
        need(op, s.size() >= 2, "invalid shape of input.");
        const long d = s.back(), T = s[s.size() - 2];
        need(op, d % 2 == 0 && V(cs).numel() == T * d && V(sn).numel() == T * d && !V(cs).batched && !V(sn).batched, "cos / sin must be [.., T, d] tables shared by every head.");
        {   // (leading dims of the tables must be 1: one table row per token)
            const Shape cshape = V(cs).shape;
            need(op, cshape.size() >= 2 && cshape.back() == d && cshape[cshape.size() - 2] == T, "invalid shape of the cos / sin tables.");
        }
        const int y = out_val(op, s, Lay::plain, V(x).batched);
        const long bh = P.total_elems(x) / (T * d);
        P.add_step("RoPE " + op.m_name, {x, cs, sn}, {y}, [=, this] { be.check(be.api.osg_rope(be.ctx, OSG_F16, P.ptr(x), P.ptr(cs), P.ptr(sn), P.ptr(y), bh, T, (int)d), "RoPE"); });
    }

    // Expand (reference :7154-7230): numpy-style broadcast of the input to `shape` (a plan-time int64 vector).  On the device: x * ones, where
    // `ones` has the target extent in every dimension the input stretches (x * 1 is exact for every f16 value, the sign of zero included)
    // Expand [1,Hkv,1,S,d] -> [1,Hkv,rep,S,d] whose result reaches nothing but the key / value input of ScaledDotProductAttention ops, through Reshapes
    // (to [1,Hkv*rep,S,d]): the transformers `repeat_kv` of a grouped-query model.  Such an Expand is not launched (lower_sdpa reads its source).
    bool repeat_kv_only(const Operation& op, const Shape& xs, const Shape& os) {
        if (!P.sdp_attn || os.size() != 5 || xs.size() != 5 || os[0] != 1 || xs[2] != 1 || os[2] < 2) return false;
        for (int k : {0, 1, 3, 4})
            if (xs[k] != os[k]) return false;
        auto consumers_of = [&](const std::string& name) {
            std::vector<std::pair<int, int>> c;   // (op, input slot)
            auto slots = [&](int i) {
                for (size_t k = 0; k < ops()[i].m_input.size(); k++)
                    if (ops()[i].m_input[k].m_name == name) c.push_back({i, (int)k});
            };
            if (indexed) {
                auto it = consumers.find(name);
                if (it != consumers.end())
                    for (int i : it->second) slots(i);
            } else
                for (size_t i = 0; i < ops().size(); i++)
                    if (dead.empty() || !dead[i]) slots((int)i);
            return c;
        };
        for (auto& e : P.extra_outputs)
            if (e == op.m_output[0].m_name) return false;
        const auto c1 = consumers_of(op.m_output[0].m_name);
        if (c1.empty()) return false;
        for (auto [ri, rk] : c1) {
            const Operation& r = ops()[ri];
            if (r.m_type != "Reshape" || rk != 0 || r.m_output.size() != 1) return false;
            for (auto& e : P.extra_outputs)
                if (e == r.m_output[0].m_name) return false;
            const auto c2 = consumers_of(r.m_output[0].m_name);
            if (c2.empty()) return false;
            for (auto [si, sk] : c2)
                if (ops()[si].m_type != "ScaledDotProductAttention" || (sk != 1 && sk != 3)) return false;
        }
        return true;
    }

    void lower_expand(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        need(op, op.m_attributes.empty(), "unrecognized attribute (not implemented).");
        const Val* sv = hval(op.m_input[1]);
        need(op, sv && sv->dtype == OSG_I64 && sv->host_valid, "wrong data type of shape.");
        need(op, sv->shape.size() == 1, "shape must be 1D.");
        const int x = P.ensure_plain(in_val(op.m_input[0]));
        need(op, V(x).dtype == OSG_F16, "wrong data type of input (only the arithmetic type is supported on the device).");
        Shape xs = V(x).shape;
        Shape ts(sv->host_i.begin(), sv->host_i.end());
        need(op, xs.size() <= ts.size(), "invalid shape of input.");
        while (xs.size() < ts.size()) xs.insert(xs.begin(), 1);
        Shape os(ts.size()), on(ts.size(), 1);
        for (size_t k = 0; k < ts.size(); k++) {
            need(op, ts[k] > 0, "dimension <= 0.");
            need(op, xs[k] == ts[k] || xs[k] == 1 || ts[k] == 1, "shape of input not matching 'shape'.");
            os[k] = std::max(xs[k], ts[k]);
            if (xs[k] == 1 && ts[k] > 1) on[k] = ts[k];
        }
        const int y = out_val(op, os, Lay::plain, V(x).batched);
        if (repeat_kv_only(op, xs, os)) {   // grouped-query attention's repeat_kv: the attention kernel maps query head h to kv head h / rep itself
            V(y).rep_src = x;
            V(y).rep = os[2];
            return;
        }
        const size_t prank = os.size() + 1;
        need(op, prank <= 6, "rank too large for the device broadcast kernel.");
        const long n_ones = prod(on);
        const int ones = P.new_val("", on, OSG_F16, Lay::plain, false);
        V(ones).is_const = true;
        bool fresh;
        V(ones).dptr = P.const_alloc("", std::max<size_t>((size_t)n_ones * 2, 8), &fresh);
        std::vector<uint16_t> h((size_t)n_ones, float_to_half(1.0f));
        be.check(be.api.osg_upload_sync(be.ctx, V(ones).dptr, h.data(), h.size() * 2), "osg_upload_sync");
        std::vector<long> sa(prank, 1), sb(prank, 1);
        for (size_t k = 0; k < xs.size(); k++) { sa[k + 1] = xs[k]; sb[k + 1] = on[k]; }
        sa[0] = B(x);
        P.add_step("Expand " + op.m_name, {x, ones}, {y}, [=, this] {
            be.check(be.api.osg_binary(be.ctx, OSG_F16, OSG_BIN_MUL, P.ptr(x), sa.data(), P.ptr(ones), sb.data(), P.ptr(y), (int)prank), "Expand");
        });
    }

    void lower_unary(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        int x = in_val(op.m_input[0]);
        const osg_dtype adt = cur_up ? OSG_F32 : OSG_F16;
        need(op, V(x).dtype == adt, "wrong data type of input.");
        const std::string& t = op.m_type;
        const osg_unary_kind k = t == "Sigmoid" ? OSG_UN_SIGMOID : t == "Erf" ? OSG_UN_ERF : t == "Sqrt" ? OSG_UN_SQRT : t == "Sin" ? OSG_UN_SIN
                                 : t == "Cos" ? OSG_UN_COS : t == "Neg" ? OSG_UN_NEG : OSG_UN_SILU;
        int y = out_val(op, V(x).shape, V(x).lay, V(x).batched, adt);
        V(y).up32 = adt == OSG_F32;
        const long n = P.total_elems(x);
        P.add_step(t + " " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_unary(be.ctx, adt, k, P.ptr(x), P.ptr(y), n, 0.f), t.c_str()); });
    }

    // Pow (reference :5478-5604): scalar exponent only
    void lower_pow(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        int x = in_val(op.m_input[0]);
        float p = 0;
        need(op, const_scalar(op.m_input[1], &p) && cval(op.m_input[1])->shape.empty(), "power must be a scalar (not implemented).");
        const osg_dtype adt = cur_up ? OSG_F32 : OSG_F16;
        need(op, V(x).dtype == adt, "wrong data type of input.");
        int y = out_val(op, V(x).shape, V(x).lay, V(x).batched, adt);
        V(y).up32 = adt == OSG_F32;
        const long n = P.total_elems(x);
        P.add_step("Pow " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_unary(be.ctx, adt, OSG_UN_POW, P.ptr(x), P.ptr(y), n, p), "Pow"); });
    }

    // InstanceNormalization (reference :4788-5055): input [1,G,L]
    void lower_instance_norm(const Operation& op) {
        need(op, op.m_input.size() == 3, "wrong number of inputs.");
        int x = P.ensure_plain(in_val(op.m_input[0]));
        int sc = in_val(op.m_input[1]), bi = in_val(op.m_input[2]);
        const Shape s = V(x).shape;
        need(op, s.size() == 3 && s[0] == 1, "input shape must be [1,G,L] (not implemented).");
        need(op, V(sc).numel() == s[1] && V(bi).numel() == s[1] && V(sc).dtype == OSG_F32 && V(bi).dtype == OSG_F32, "invalid scale/bias.");
        float eps = 1e-5f;
        for (auto& a : op.m_attributes) {
            if (a.first == "epsilon") eps = std::stof(a.second);
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        int y = out_val(op, s, Lay::plain, V(x).batched);
        const long rows = s[1] * B(x), L = s[2], G = s[1];
        P.add_step("InstanceNorm " + op.m_name, {x, sc, bi}, {y}, [=, this] {
            be.check(be.api.osg_instance_norm(be.ctx, OSG_F16, P.ptr(x), (const float*)P.ptr(sc), (const float*)P.ptr(bi), P.ptr(y), (int)rows, L,
                                              (int)G, eps),
                     "InstanceNormalization");
        });
    }

    void lower_group_norm(const Operation& op) {
        int x = P.ensure_nhwc(in_val(op.m_input[0]));
        int g = in_val(op.m_input[1]), b = in_val(op.m_input[2]);
        const Shape s = V(x).shape;
        const long G = std::stol(*attr(op, "groups"));
        const float eps = std::stof(*attr(op, "epsilon"));
        const int act = *attr(op, "silu") == "1" ? OSG_ACT_SILU : OSG_ACT_NONE;
        int y = out_val(op, s, Lay::nhwc, V(x).batched);
        const long nb = B(x), HW = s[2] * s[3], C = s[1];
        // ---- statistics from the producer (round 3, m_hip_gn_stats): when x is what convolutions store -- one convolution's output, or a Concat buffer that
        // convolutions fill through output views -- their epilogues add the per-(image, group) sums up on the way out (osg_set_stat_sinks) and the
        // normalisation is ONE streaming launch that reads 2 numbers per group.  Levels with >= 2048 rows (the 64 x 64 and 32 x 32 images of the SD 1.5 UNet):
        // below that the convolutions run split-K, whose slabs have no finished values to add up, and the statistics would cost a launch of their own.
        if (P.gn_stats_on && V(x).dtype == OSG_F16 && HW % 128 == 0 && nb * HW >= 2048 && nb * HW * C >= P.gn_stats_min_elems && C % 8 == 0 && C % G == 0 && V(x).ld == 0 && V(x).view_off == 0) {
            std::vector<ConcatPart> parts;
            const int rx = P.root_of(x);
            auto cp = concat_parts.find(rx);
            if (cp != concat_parts.end()) parts = cp->second;      // (every operand of the Concat is stored there by a convolution: together they cover all channels)
            else {
                auto it = conv_producers.find(rx);
                if (it != conv_producers.end() && it->second.y == x && it->second.out->dst == x && it->second.out->dst_ld == 0) parts.push_back(ConcatPart{it->second.out, 0, 0, it->second.step});
            }
            bool ok = !parts.empty();
            for (auto& pt : parts) ok = ok && !pt.out->no_sinks && pt.out->sink[pt.slot].off < 0 && (pt.out->sink_hw == 0 || pt.out->sink_hw == (int)HW);
            if (ok) {
                const long off = P.gn_stats_bytes;
                P.gn_stats_bytes += (size_t)8 * nb * G * 2 * sizeof(long long);     // (eight copies, one per XCD: include/osgpu.h osg_set_stat_sinks)
                for (auto& pt : parts) {
                    pt.out->sink[pt.slot] = StatSinkRef{off, (int)G, (int)(C / G), (int)pt.ch_off};
                    pt.out->sink_hw = (int)HW;
                    P.steps[pt.step].what += " +gnstats";
                }
                P.add_step("GroupNorm stats< " + op.m_name, {x, g, b}, {y}, [=, this] {
                    be.check(be.api.osg_group_norm_stats_nhwc(be.ctx, P.ptr(x), P.ptr(g), P.ptr(b), P.ptr(y), (int)nb, HW, (int)C, (int)G, eps, (osg_act)act, P.gn_stats + off),
                             "GroupNorm");
                });
                return;
            }
        }
        P.add_step("GroupNorm " + op.m_name, {x, g, b}, {y}, [=, this] {
            be.check(be.api.osg_group_norm_nhwc(be.ctx, OSG_F16, P.ptr(x), P.ptr(g), P.ptr(b), P.ptr(y), (int)nb, HW, (int)C, (int)G, eps, (osg_act)act),
                     "GroupNorm");
        });
    }

    // ---- LayerNorm folded into the GEMM(s) that consume it (osg_gemm_ln): when every consumer of a LayerNorm is an osg.Linear that ends
    // up in ONE GEMM launch (a single Linear / Linear+GEGLU, or the members of one merged group: self-attention Q|K|V), the LayerNorm is
    // not launched at all -- gamma moves into the weight, beta and the mean correction into two fp32 epilogue vectors, the row
    // statistics are accumulated by the GEMM's math waves beside the MFMAs, from the A fragments they read anyway.  48 launches less in the SD 1.5 UNet.
    std::map<std::string, LnFold> ln_deferred;   // LayerNorm output name -> what its consumers fold
    // ConstPool tag of a folded copy: it must name everything that went into the fold -- the LayerNorm's gamma and beta and the Linear's bias -- or a
    // weight tensor shared by two Linears behind different LayerNorms / with different biases would silently reuse the first fold (advisor, round 2)
    std::string ln_tag(const LnFold& f, int bias) {
        return "|ln:" + V(f.g).name + ":" + V(f.b).name + ":" + (bias >= 0 ? V(bias).name : std::string("-"));
    }

    bool ln_can_fold(const Operation& op, int x, int g, int b, long C) {
        if (P.fusion < 2 || !P.fuse_ln_gemm || P.stream_weights) return false;
        if (C % 64) return false;
        if (!V(g).host_valid || !V(b).host_valid || (long)V(g).host_f.size() != C || (long)V(b).host_f.size() != C) return false;
        if (V(x).ld != 0) return false;
        const std::string& out = op.m_output[0].m_name;
        if (use_count(out) >= 1000) return false;
        auto cit = consumers.find(out);
        if (cit == consumers.end() || cit->second.empty()) return false;
        int group = -2;
        for (int c : cit->second) {
            const Operation& co = ops()[c];
            if (dead[c] || co.m_type != "osg.Linear" || co.m_input.empty() || co.m_input[0].m_name != out) return false;
            for (size_t k = 1; k < co.m_input.size(); k++)
                if (co.m_input[k].m_name == out) return false;
            const Val* w = cval(co.m_input[1]);
            if (!w || w->shape.size() != 2 || w->dtype != OSG_F16 || w->shape[0] != C || w->shape[1] % 4) return false;
            if (co.m_input.size() > 2 && !co.m_input[2].m_name.empty() && (!cval(co.m_input[2]) || cval(co.m_input[2])->dtype != OSG_F16)) return false;
            auto git = group_of.find(c);
            const int gi = git == group_of.end() ? -1 : git->second.first;
            if (group == -2) group = gi;
            else if (group != gi || gi < 0) return false;         // several consumers must be members of ONE merged group
        }
        if (group >= 0 && groups[group].members.size() != cit->second.size()) return false;   // ... and be all of its members
        return true;
    }

    // W[N,K] (device, f16) -> W' = f16(gamma[k] * W[n][k]) in place; returns the fp32 device vectors (c1, c2):
    // c1[n] = sum_k W'[n][k], c2[n] = sum_k beta[k] * W[n][k] + bias[n]
    std::pair<int, int> ln_fold_weight(const LnFold& f, int wnk, int bias, void* w_dst, const std::string& tag) {
        const long Nn = V(wnk).shape[0], K = V(wnk).shape[1];
        bool fresh1, fresh2;
        int v1 = P.new_val("", {Nn}, OSG_F32, Lay::plain, false), v2 = P.new_val("", {Nn}, OSG_F32, Lay::plain, false);
        V(v1).is_const = V(v2).is_const = true;
        V(v1).dptr = P.const_alloc(tag + "|c1", (size_t)Nn * 4, &fresh1);
        V(v2).dptr = P.const_alloc(tag + "|c2", (size_t)Nn * 4, &fresh2);
        if (!fresh1 && !fresh2) return {v1, v2};   // folded by an earlier plan of this Model (w_dst is the same pooled buffer)
        std::vector<uint16_t> w((size_t)Nn * K);
        be.check(be.api.osg_sync(be.ctx), "osg_sync");
        be.check(be.api.osg_download(be.ctx, w.data(), P.ptr(wnk), w.size() * 2), "osg_download");
        std::vector<uint16_t> bh;
        if (bias >= 0) {
            bh.resize((size_t)Nn);
            be.check(be.api.osg_download(be.ctx, bh.data(), P.ptr(bias), bh.size() * 2), "osg_download");
        }
        const std::vector<float>& gam = V(f.g).host_f;
        const std::vector<float>& bet = V(f.b).host_f;
        std::vector<float> c1((size_t)Nn), c2((size_t)Nn);
        for (long n = 0; n < Nn; n++) {
            double s1 = 0, s2 = 0;
            uint16_t* row = w.data() + (size_t)n * K;
            for (long k = 0; k < K; k++) {
                const float wv = half_to_float(row[k]);
                s2 += (double)bet[k] * (double)wv;
                const uint16_t folded = float_to_half(gam[k] * wv);
                row[k] = folded;
                s1 += (double)half_to_float(folded);
            }
            c1[n] = (float)s1;
            c2[n] = (float)(s2 + (bias >= 0 ? (double)half_to_float(bh[n]) : 0.0));
        }
        be.check(be.api.osg_upload_sync(be.ctx, w_dst, w.data(), w.size() * 2), "osg_upload_sync");
        be.check(be.api.osg_upload_sync(be.ctx, V(v1).dptr, c1.data(), c1.size() * 4), "osg_upload_sync");
        be.check(be.api.osg_upload_sync(be.ctx, V(v2).dptr, c2.data(), c2.size() * 4), "osg_upload_sync");
        return {v1, v2};
    }

    // a private [N,K] copy of a resident weight (the K-contiguous twin is shared by every user of the tensor: never folded in place)
    int private_copy(int wnk, const std::string& tag) {
        const long Nn = V(wnk).shape[0], K = V(wnk).shape[1];
        int t = P.new_val("", {Nn, K}, OSG_F16, Lay::plain, false);
        V(t).is_const = true;
        V(t).name = V(wnk).name + tag;
        bool fresh;
        V(t).dptr = P.const_alloc(V(t).name, (size_t)Nn * K * 2, &fresh);
        return t;
    }

    void emit_gemm_ln(const std::string& what, const LnFold& f, int wfold, int c1, int c2, int res, int y, long M, long Nn, long K, osg_act act_) {
        const int x = f.x, rs = f.rs;
        const float eps = f.eps;
        std::vector<int> reads = {x, wfold, c1, c2};
        if (res >= 0) reads.push_back(res);
        if (rs >= 0) reads.push_back(rs);
        P.add_step(what, reads, {y}, [=, this] {
            be.check(be.api.osg_gemm_ln(be.ctx, P.ptr(x), P.ptr(wfold), (const float*)P.ptr(c1), (const float*)P.ptr(c2), eps,
                                        rs >= 0 ? (const float*)P.ptr(rs) : nullptr, res >= 0 ? P.ptr(res) : nullptr, P.ptr(y), (int)M, (int)Nn, (int)K, act_),
                     what.c_str());
        });
        P.steps.back().flops = 2.0 * M * Nn * K;
    }

    void lower_layer_norm(const Operation& op) {
        int x = P.ensure_plain(in_val(op.m_input[0]));
        int g = in_val(op.m_input[1]), b = in_val(op.m_input[2]);
        const Shape s = V(x).shape;
        const float eps = std::stof(*attr(op, "epsilon"));
        if (ln_can_fold(op, x, g, b, s.back())) {
            check_out(op, s);
            const int rs = upgrade_rs_producer(x, P.total_elems(x) / s.back(), s.back());
            ln_deferred[op.m_output[0].m_name] = LnFold{x, g, b, eps, s.back(), rs};
            return;
        }
        int y = out_val(op, s, Lay::plain, V(x).batched);
        const long C = s.back(), rows = P.total_elems(x) / C;
        P.add_step("LayerNorm " + op.m_name, {x, g, b}, {y}, [=, this] {
            be.check(be.api.osg_layer_norm(be.ctx, OSG_F16, P.ptr(x), P.ptr(g), P.ptr(b), P.ptr(y), rows, (int)C, eps), "LayerNorm");
        });
    }

    void lower_geglu(const Operation& op) {
        int x = P.ensure_plain(in_val(op.m_input[0]));
        Shape s = V(x).shape;
        const long C = s.back() / 2, rows = P.total_elems(x) / s.back();
        s.back() = C;
        int y = out_val(op, s, Lay::plain, V(x).batched);
        P.add_step("GEGLU " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_geglu(be.ctx, OSG_F16, P.ptr(x), P.ptr(y), rows, C), "GEGLU"); });
    }

    void lower_attention(const Operation& op) {
        int q = P.ensure_plain(in_val_raw(op.m_input[0])), k = P.ensure_plain(in_val_raw(op.m_input[1])), v = P.ensure_plain(in_val_raw(op.m_input[2]));
        const long h = std::stol(*attr(op, "heads"));
        const float scale = std::stof(*attr(op, "scale"));
        const Shape qs = V(q).shape;
        const long Tq = qs[1], C = qs[2], Tk = V(k).shape[1], d = C / h;
        int y = out_val(op, qs, Lay::plain, V(q).batched);
        need(op, V(q).batched == V(k).batched && V(k).batched == V(v).batched, "q/k/v batching mismatch.");
        const long nb = B(q);
        P.add_step("Attention " + op.m_name, {q, k, v}, {y}, [=, this] {
            const long lq = P.vals[q].ld ? P.vals[q].ld : C, lk = P.vals[k].ld ? P.vals[k].ld : C, lv = P.vals[v].ld ? P.vals[v].ld : C;
            be.check(be.api.osg_attention_strided(be.ctx, OSG_F16, P.ptr(q), lq, d, Tq * lq, P.ptr(k), lk, d, Tk * lk, P.ptr(v), lv, d, Tk * lv,
                                                  P.ptr(y), C, d, Tq * C, (int)nb, (int)h, (int)Tq, (int)Tk, (int)d, scale),
                     "Attention");
        });
        P.steps.back().flops = 4.0 * nb * h * Tq * Tk * d;
    }

    // ---- osg.TBlockTail (fuse_tblock_tail): one launch of osg_tblock_tail per transformer block, K / V packs of ALL such blocks from one launch --------
    std::map<std::string, std::pair<int, size_t>> kv_packs;   // fused op name -> (pack buffer val, element offset of its kp; vtp right behind)

    // K / V of cross-attentions that are column views of ONE merged projection of the text context (plan_linear_groups: 32 -> 1) are re-packed by ONE
    // launch for every osg.TBlockTail of the graph that shares the merged output and the head count with `op`
    // (input positions of k / v in the fused ops that read packs)
    static int kv_input(const Operation& o) { return o.m_type == "osg.TBlockTail" ? 8 : -1; }
    void prepack_kv(const Operation& op, int k, int v, long heads, long Tk) {
        struct Job { std::string name; long kcol, vcol, D; };
        const int ki0 = kv_input(op);
        std::vector<Job> jobs;
        const int base = P.root_of(k);
        const long ld = V(k).ld;
        const bool shared = ld > 0 && P.root_of(v) == base && V(v).ld == ld && V(k).view_off % 2 == 0 && V(v).view_off % 2 == 0;
        const long imgs = B(k);
        int src = base;
        long src_ld = ld;
        if (shared) {
            jobs.push_back(Job{op.m_name, (long)V(k).view_off / 2, (long)V(v).view_off / 2, V(k).shape.back() / heads});
            const int gk = indexed && producer.count(op.m_input[ki0].m_name) && group_of.count(producer[op.m_input[ki0].m_name]) ? group_of[producer[op.m_input[ki0].m_name]].first : -1;
            if (gk >= 0)
                for (size_t i = 0; i < ops().size(); i++) {
                    const Operation& o = ops()[i];
                    const int ki = kv_input(o);
                    if (ki < 0 || o.m_name == op.m_name || kv_packs.count(o.m_name) || std::stol(*attr(o, "heads")) != heads) continue;
                    auto pk = producer.find(o.m_input[ki].m_name), pv = producer.find(o.m_input[ki + 1].m_name);
                    if (pk == producer.end() || pv == producer.end()) continue;
                    auto gik = group_of.find(pk->second), giv = group_of.find(pv->second);
                    if (gik == group_of.end() || giv == group_of.end() || gik->second.first != gk || giv->second.first != gk) continue;
                    if ((long)o.m_input[ki].m_shape[1] != Tk) continue;
                    const LinGroup& g = groups[gk];
                    jobs.push_back(Job{o.m_name, g.off[gik->second.second], g.off[giv->second.second], (long)o.m_input[ki].m_shape[2] / heads});
                }
        } else {
            // dense K and V of one block: side by side in a scratch matrix first (never the case in the SD graphs: their K / V projections are merged)
            const long C = V(k).shape.back();
            const int kd = P.ensure_plain(k), vd = P.ensure_plain(v);
            src = P.new_val("", {imgs * Tk, 2 * C}, OSG_F16, Lay::plain, false);
            src_ld = 2 * C;
            const long rows = imgs * Tk;
            P.add_step("KVPack gather " + op.m_name, {kd, vd}, {src}, [=, this] { be.check(be.api.osg_concat2(be.ctx, 2, P.ptr(kd), C, P.ptr(vd), C, P.ptr(src), rows), "KVPack"); });
            jobs.push_back(Job{op.m_name, 0, C, C / heads});
        }
        std::vector<int> table;
        size_t total = 0;
        for (auto& j : jobs) {
            const size_t each = be.api.osg_tblock_kv_pack_elems((int)imgs, (int)heads, (int)j.D);
            table.insert(table.end(), {(int)j.kcol, (int)j.vcol, (int)j.D, (int)total});
            total += 2 * each;
        }
        const int pack = P.new_val("", {(long)total}, OSG_F16, Lay::plain, false);
        void* tdev = P.small_alloc(table.size() * sizeof(int));
        be.check(be.api.osg_upload_sync(be.ctx, tdev, table.data(), table.size() * sizeof(int)), "osg_upload_sync");
        for (size_t j = 0; j < jobs.size(); j++) kv_packs[jobs[j].name] = {pack, (size_t)table[4 * j + 3]};
        const int nj = (int)jobs.size();
        P.add_step("KVPack x" + std::to_string(nj) + " " + op.m_name, {src}, {pack}, [=, this] {
            be.check(be.api.osg_tblock_kv_pack_jobs(be.ctx, P.ptr(src), src_ld, (int)imgs, (int)Tk, (int)heads, nj, (const int*)tdev, P.ptr(pack)), "KVPack");
        });
    }

    // a resident [N,K] weight in the layout osg_tblock_tail streams ([K/8][N][8], osg_tblock_pack_weight): a derived constant of the Model's pool, made once
    int weight_kn8(int wnk) {
        const long Nn = V(wnk).shape[0], K = V(wnk).shape.size() == 2 ? V(wnk).shape[1] : prod(V(wnk).shape) / Nn;
        int t = P.new_val("", {K / 8, Nn, 8}, OSG_F16, Lay::plain, false);
        V(t).is_const = true;
        V(t).name = V(wnk).name + "|kn8";
        bool fresh;
        V(t).dptr = P.const_alloc(V(t).name, (size_t)Nn * K * 2, &fresh);
        if (fresh) {
            be.check(be.api.osg_tblock_pack_weight(be.ctx, P.ptr(wnk), (int)Nn, (int)K, V(t).dptr), "osg_tblock_pack_weight");
            be.check(be.api.osg_sync(be.ctx), "osg_sync");
        }
        return t;
    }

    void lower_tblock_tail(const Operation& op) {
        const bool proj = *attr(op, "proj") == "1";
        need(op, op.m_input.size() == (proj ? 21u : 18u) && op.m_output.size() == 1, "wrong number of inputs.");
        const long heads = std::stol(*attr(op, "heads"));
        const float scale = std::stof(*attr(op, "scale")), eps2 = std::stof(*attr(op, "eps2")), eps3 = std::stof(*attr(op, "eps3"));
        const int a1 = P.ensure_plain(in_val(op.m_input[0])), x0 = P.ensure_plain(in_val(op.m_input[1]));
        auto opt = [&](const Tensor& t) { return t.m_name.empty() ? -1 : in_val(t); };
        const int wo1 = weight_kn8(weight_nk(in_val(op.m_input[2]))), bo1 = opt(op.m_input[3]), g2 = in_val(op.m_input[4]), be2 = in_val(op.m_input[5]);
        const int wq2 = weight_kn8(weight_nk(in_val(op.m_input[6]))), bq2 = opt(op.m_input[7]);
        const int k = in_val_raw(op.m_input[8]), v = in_val_raw(op.m_input[9]);
        const int wo2 = weight_kn8(weight_nk(in_val(op.m_input[10]))), bo2 = opt(op.m_input[11]), g3 = in_val(op.m_input[12]), be3 = in_val(op.m_input[13]);
        const int w1 = weight_kn8(weight_nk(in_val(op.m_input[14]))), b1 = opt(op.m_input[15]), w2 = weight_kn8(weight_nk(in_val(op.m_input[16]))), b2 = opt(op.m_input[17]);
        const Shape as = V(a1).shape;
        need(op, as.size() == 3 && as[0] == 1 && V(x0).shape == as && V(a1).ld == 0 && V(x0).ld == 0, "invalid shape of inputs.");
        const long T = as[1], C = as[2], F = 4 * C, Tk = V(k).shape[1], nb = B(a1), M = T * nb;
        need(op, V(a1).batched == V(x0).batched && V(k).batched == V(a1).batched && V(v).batched == V(a1).batched, "q/k/v batching mismatch.");
        need(op, be.api.osg_tblock_tail_supported((int)M, (int)T, (int)C, (int)heads, (int)Tk) == 1, "shape not taken by osg_tblock_tail.");
        for (int g : {g2, be2, g3, be3}) need(op, V(g).dtype == OSG_F16 && V(g).numel() == C, "invalid LayerNorm operands.");
        int wpo = -1, bpo = -1, xin = -1;
        if (proj) {
            wpo = in_val(op.m_input[18]);
            bpo = opt(op.m_input[19]);
            xin = P.ensure_nhwc(in_val(op.m_input[20]));
            need(op, V(wpo).is_const && V(wpo).lay == Lay::nhwc && V(wpo).dtype == OSG_F16 && V(xin).ld == 0 && P.total_elems(xin) == M * C, "invalid proj_out operands.");
            wpo = weight_kn8(wpo);      // (OHWI of a 1x1 convolution IS [N][K])
        }
        if (!kv_packs.count(op.m_name)) prepack_kv(op, k, v, heads, Tk);
        const auto [pack, koff] = kv_packs.at(op.m_name);
        const size_t each = be.api.osg_tblock_kv_pack_elems((int)nb, (int)heads, (int)(C / heads));
        int y;
        if (proj) {
            const Shape xs = V(xin).shape;   // [1, C, H, W]
            y = out_val(op, xs, Lay::nhwc, V(a1).batched);
        } else
            y = out_val(op, as, Lay::plain, V(a1).batched);
        auto co = std::make_shared<ConvOut>();
        co->dst = y;
        co->no_sinks = true;
        std::vector<int> reads = {a1, x0, wo1, g2, be2, wq2, pack, wo2, g3, be3, w1, w2};
        for (int r : {bo1, bq2, bo2, b1, b2, wpo, bpo, xin})
            if (r >= 0) reads.push_back(r);
        const std::string what = std::string("TBlockTail") + (proj ? "+proj_out " : " ") + op.m_name;
        P.add_step(what, reads, {y}, [=, this] {
            const ConvOut& o = *co;
            auto p = [&](int val) -> const void* { return val >= 0 ? P.ptr(val) : nullptr; };
            osg_tblock_tail_args a{};
            a.a1 = p(a1); a.x0 = p(x0);
            a.wo1 = p(wo1); a.bo1 = p(bo1);
            a.g2 = p(g2); a.be2 = p(be2); a.eps2 = eps2;
            a.wq2 = p(wq2); a.bq2 = p(bq2);
            a.kp = (const char*)P.ptr(pack) + koff * 2;
            a.vtp = (const char*)a.kp + each * 2;
            a.scale = scale; a.Tk = (int)Tk;
            a.wo2 = p(wo2); a.bo2 = p(bo2);
            a.g3 = p(g3); a.be3 = p(be3); a.eps3 = eps3;
            a.w1 = p(w1); a.b1 = p(b1); a.w2 = p(w2); a.b2 = p(b2);
            a.wpo = p(wpo); a.bpo = p(bpo); a.xin = p(xin);
            a.out = (char*)P.ptr(o.dst) + o.dst_off; a.ldo = o.dst_ld;
            a.out2 = o.dst2 >= 0 ? (char*)P.ptr(o.dst2) + o.dst2_off : nullptr; a.ldo2 = o.dst2_ld;
            a.M = (int)M; a.rows_per_img = (int)T; a.C = (int)C; a.heads = (int)heads;
            be.check(be.api.osg_tblock_tail(be.ctx, &a), what.c_str());
        });
        P.steps.back().flops = 2.0 * M * C * C * (proj ? 4 : 3) + 2.0 * M * C * 2 * F + 2.0 * M * F * C + 4.0 * nb * heads * T * Tk * (C / heads);
        if (proj) conv_producers[P.root_of(y)] = ConvProducer{P.steps.size() - 1, co, y};
    }

    // ScaledDotProductAttention (reference :7767-7882): q [B,Hq,T,D], k [B,Hkv,S,D], mask [T,S] | [1,1,T,S], v [B,Hkv,S,Dv] -> [B,Hq,T,Dv]
    void lower_sdpa(const Operation& op) {
        need(op, op.m_input.size() == 4, "wrong number of inputs.");
        need(op, op.m_output.size() == 1, "wrong number of outputs.");
        // key / value: a virtual repeat_kv Expand (lower_expand) is read at its source, with the un-repeated head count
        auto kv_operand = [&](const Tensor& t) {
            const int v0 = in_val(t), r = P.root_of(v0);
            if (V(r).rep_src < 0) return P.ensure_plain(v0);
            const int src = P.ensure_plain(V(r).rep_src);
            const Shape ss = V(src).shape, es = V(v0).shape;   // [1,Hkv,1,S,d] and [1,Hkv*rep,S,d]
            need(op, es.size() == 4 && ss.size() == 5 && es[0] == 1 && es[1] == ss[1] * V(r).rep && es[2] == ss[3] && es[3] == ss[4], "invalid shape of key or value.");
            return P.alias(src, Shape{1, ss[1], ss[3], ss[4]}, Lay::plain);
        };
        const int q = P.ensure_plain(in_val(op.m_input[0])), k = kv_operand(op.m_input[1]), mk = P.ensure_plain(in_val(op.m_input[2])), v = kv_operand(op.m_input[3]);
        const Shape qs = V(q).shape, ks = V(k).shape, vs = V(v).shape, ms = V(mk).shape;
        need(op, qs.size() == 4, "invalid shape of query.");
        need(op, ks.size() == 4, "invalid shape of key.");
        need(op, vs.size() == 4, "invalid shape of value.");
        need(op, ms.size() == 2 || (ms.size() == 4 && ms[0] == 1 && ms[1] == 1), "invalid shape of mask.");
        const long Bq = qs[0], Hq = qs[1], T = qs[2], D = qs[3], Hkv = ks[1], S = ks[2], Dv = vs[3];
        need(op, ks[0] == Bq && vs[0] == Bq && ks[3] == D && vs[1] == Hkv && vs[2] == S && Hkv > 0 && Hq % Hkv == 0, "invalid shape of query, key or value.");
        need(op, ms[ms.size() - 2] == T && ms[ms.size() - 1] == S, "invalid shape of mask.");
        need(op, V(q).dtype == OSG_F16 && V(k).dtype == OSG_F16 && V(v).dtype == OSG_F16 && V(mk).dtype == OSG_F16, "wrong data type of query.");
        need(op, Dv == D && D % 8 == 0 && D <= 160, "head dims other than a multiple of 8 up to 160 with Dv == D are not implemented on the HIP backend.");
        need(op, N == 1 || (!V(mk).batched && V(q).batched == V(k).batched && V(k).batched == V(v).batched), "q/k/v batching mismatch.");
        const float scale = std::stof(*attr(op, "scale"));
        need(op, scale > 0.f, "a scale <= 0 is not implemented on the HIP backend.");
        const int y = out_val(op, Shape{Bq, Hq, T, Dv}, Lay::plain, V(q).batched);
        const long nb = Bq * B(q);
        P.add_step("ScaledDotProductAttention " + op.m_name, {q, k, mk, v}, {y}, [=, this] {
            be.check(be.api.osg_sdpa(be.ctx, OSG_F16, P.ptr(q), P.ptr(k), P.ptr(v), P.ptr(mk), P.ptr(y), (int)nb, (int)Hq, (int)Hkv, (int)T, (int)S, (int)D, scale),
                     "ScaledDotProductAttention");
        });
        P.steps.back().flops = 4.0 * nb * Hq * T * S * D;
    }

    // AttentionFusedOps (reference :6696-6929): q [n,Tq,d], k [n,d,Tk] (already transposed), optional scalar s, v [n,Tk,d]
    void lower_attention_fused_ops(const Operation& op) {
        need(op, op.m_input.size() == 4, "wrong number of inputs.");
        int q = P.ensure_plain(in_val(op.m_input[0])), k = P.ensure_plain(in_val(op.m_input[1])), v = P.ensure_plain(in_val(op.m_input[3]));
        Shape qs = V(q).shape, ks = V(k).shape, vs = V(v).shape;
        bool lead1 = false;
        if (qs.size() == 4 && qs[0] == 1 && ks.size() == 4 && ks[0] == 1 && vs.size() == 4 && vs[0] == 1) {
            qs.erase(qs.begin()); ks.erase(ks.begin()); vs.erase(vs.begin());
            lead1 = true;
        }
        need(op, qs.size() == 3 && ks.size() == 3 && vs.size() == 3, "shapes of q, k and v must have 3 dimensions.");
        need(op, qs[0] == ks[0] && qs[0] == vs[0] && ks[1] == qs[2] && vs[1] == ks[2] && vs[2] == qs[2], "invalid shape(s) of q, k and/or v.");
        float scale = 1.0f;
        if (!op.m_input[2].m_name.empty()) {
            need(op, cval(op.m_input[2]) && cval(op.m_input[2])->shape.empty() && const_scalar(op.m_input[2], &scale), "s must be a scalar.");
        }
        need(op, (size_t)qs[1] >= m.m_attention_fused_ops_parts, "m_attention_fused_ops_parts is not valid.");
        Shape os = qs;
        if (lead1) os.insert(os.begin(), 1);
        int y = out_val(op, os, Lay::plain, V(q).batched);
        need(op, V(q).batched == V(k).batched && V(k).batched == V(v).batched, "q/k/v batching mismatch.");
        const long heads = qs[0] * B(q), Tq = qs[1], d = qs[2], Tk = ks[2];
        if (d > 160 || d % 8) {
            // head dims the flash kernel does not take (the VAE's single 512-wide head): the reference's own sequence, unsliced --
            // S = f16(Q K^T); S = f16(S * s); P = softmax_rows(S); O = f16(P V)  (reference :6796-6929)
            int sv = P.new_val("", {qs[0], Tq, Tk}, OSG_F16, Lay::plain, V(q).batched);
            int pv = P.new_val("", {qs[0], Tq, Tk}, OSG_F16, Lay::plain, V(q).batched);
            emit_gemm("MatMul " + op.m_name + "/QK", q, k, -1, -1, sv, Tq, Tk, d, heads, Tq * d, d * Tk, Tq * Tk, 0);
            int cur = sv;
            if (scale != 1.0f) {
                int sc = P.new_val("", {1}, OSG_F16, Lay::plain, false);
                V(sc).is_const = true;
                bool fresh;
                V(sc).dptr = P.const_alloc("scale|" + op.m_name, 256, &fresh);
                const uint16_t hbits = float_to_half(scale);
                if (fresh) be.check(be.api.osg_upload_sync(be.ctx, V(sc).dptr, &hbits, 2), "osg_upload_sync");
                const long tot = heads * Tq * Tk;
                P.add_step("Mul " + op.m_name + "/scale", {sv, sc}, {sv}, [=, this] {
                    long as[1] = {tot}, bs[1] = {1};
                    be.check(be.api.osg_binary(be.ctx, OSG_F16, OSG_BIN_MUL, P.ptr(sv), as, P.ptr(sc), bs, P.ptr(sv), 1), "osg_binary");
                });
            }
            P.add_step("Softmax " + op.m_name, {cur}, {pv}, [=, this] {
                be.check(be.api.osg_softmax_last(be.ctx, OSG_F16, P.ptr(cur), P.ptr(pv), heads * Tq, Tk), "osg_softmax_last");
            });
            emit_gemm("MatMul " + op.m_name + "/PV", pv, v, -1, -1, y, Tq, d, Tk, heads, Tq * Tk, Tk * d, Tq * d, 0);
            return;
        }
        P.add_step("AttentionFusedOps " + op.m_name, {q, k, v}, {y}, [=, this] {
            be.check(be.api.osg_attention(be.ctx, OSG_F16, P.ptr(q), P.ptr(k), P.ptr(v), P.ptr(y), (int)heads, (int)Tq, (int)Tk, (int)d, scale, 1),
                     "AttentionFusedOps");
        });
        P.steps.back().flops = 4.0 * heads * Tq * Tk * d;
    }

    // ReduceMean (reference :5237-5393): last axis, keepdims
    void lower_reduce_mean(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        int x = P.ensure_plain(in_val(op.m_input[0]));
        const Shape s = V(x).shape;
        for (auto& a : op.m_attributes) {
            if (a.first == "axes") {
                auto ax = int_list(a.second);
                need(op, ax.size() == 1 && (ax[0] == -1 || ax[0] == (int)s.size() - 1), "reduction supported on the last axis only (not implemented).");
            } else if (a.first == "keepdims") need(op, std::stoi(a.second) == 1, "keepdims must be 1 (not implemented).");
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        Shape os = s;
        os.back() = 1;
        const osg_dtype adt = cur_up ? OSG_F32 : OSG_F16;
        need(op, V(x).dtype == adt, "wrong data type of input.");
        int y = out_val(op, os, Lay::plain, V(x).batched, adt);
        V(y).up32 = adt == OSG_F32;
        const long C = s.back(), rows = P.total_elems(x) / C;
        P.add_step("ReduceMean " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_reduce_mean_last(be.ctx, adt, P.ptr(x), P.ptr(y), rows, C), "ReduceMean"); });
    }

    // Softmax (reference :5862-5998): any axis via transpose-in / softmax / transpose-out (:5883-5923)
    void lower_softmax(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        int x = P.ensure_plain(in_val(op.m_input[0]));
        const Shape s = V(x).shape;
        int axis = -1;
        for (auto& a : op.m_attributes) {
            if (a.first == "axis") axis = std::stoi(a.second);
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        const int rank = (int)s.size();
        if (axis < 0) axis += rank;
        need(op, axis >= 0 && axis < rank, "invalid axis.");
        int y = out_val(op, s, Lay::plain, V(x).batched);
        const long nb = B(x);
        if (axis == rank - 1) {
            const long C = s.back(), rows = P.total_elems(x) / C;
            P.add_step("Softmax " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_softmax_last(be.ctx, OSG_F16, P.ptr(x), P.ptr(y), rows, C), "Softmax"); });
            return;
        }
        // [outer, A, inner] -> [outer, inner, A] -> softmax -> back
        const long outer = prod(s, 0, axis) * nb, A = s[axis], inner = prod(s, axis + 1);
        int t0 = P.new_val("", s, OSG_F16, Lay::plain, V(x).batched), t1 = P.new_val("", s, OSG_F16, Lay::plain, V(x).batched);
        P.add_step("Softmax/T0 " + op.m_name, {x}, {t0}, [=, this] {
            long sh[3] = {outer, A, inner}; int pm[3] = {0, 2, 1};
            be.check(be.api.osg_transpose(be.ctx, 2, P.ptr(x), P.ptr(t0), 3, sh, pm), "Softmax");
        });
        P.add_step("Softmax " + op.m_name, {t0}, {t1}, [=, this] { be.check(be.api.osg_softmax_last(be.ctx, OSG_F16, P.ptr(t0), P.ptr(t1), outer * inner, A), "Softmax"); });
        P.add_step("Softmax/T1 " + op.m_name, {t1}, {y}, [=, this] {
            long sh[3] = {outer, inner, A}; int pm[3] = {0, 2, 1};
            be.check(be.api.osg_transpose(be.ctx, 2, P.ptr(t1), P.ptr(y), 3, sh, pm), "Softmax");
        });
    }

    // Reshape (reference :4708-4787): zero-copy; an NHWC producer is brought back to the logical layout first (:4724)
    void lower_reshape(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        for (auto& a : op.m_attributes) {
            if (a.first == "allowzero") need(op, std::stoi(a.second) == 0, "allowzero != 0 not supported (not implemented).");
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        int x = in_val(op.m_input[0]);
        const Val* sv = hval(op.m_input[1]);
        need(op, sv && sv->dtype == OSG_I64 && sv->host_valid, "wrong data type of shape.");
        const long total = V(x).numel();
        Shape os;
        long known = 1;
        int infer = -1;
        for (size_t i = 0; i < sv->host_i.size(); i++) {
            long d = sv->host_i[i];
            if (d == 0) { need(op, i < V(x).shape.size(), "invalid 0 in shape."); d = V(x).shape[i]; }
            if (d == -1) { need(op, infer < 0, "more than one -1 in shape."); infer = (int)i; os.push_back(1); continue; }
            os.push_back(d);
            known *= d;
        }
        if (infer >= 0) { need(op, known && total % known == 0, "invalid shape."); os[infer] = total / known; }
        need(op, prod(os) == total, "invalid shape.");
        // layout-preserving special case: an NHWC [1,C,H,W] tensor reshaped to [1,C,H*W]... is NOT a no-op; go plain.
        x = P.ensure_plain(x);
        check_out(op, os);
        P.alias(x, os, Lay::plain, op.m_output[0].m_name);
    }

    void lower_flatten(const Operation& op) {
        int x = P.ensure_plain(in_val(op.m_input[0]));
        int axis = 1;
        if (auto* a = attr(op, "axis")) axis = std::stoi(*a);
        const Shape s = V(x).shape;
        if (axis < 0) axis += (int)s.size();
        Shape os = {prod(s, 0, axis), prod(s, axis)};
        check_out(op, os);
        P.alias(x, os, Lay::plain, op.m_output[0].m_name);
    }

    // Unsqueeze (reference :3859-3905) / Squeeze (:7425): axes from a static int64 tensor
    void lower_squeeze(const Operation& op, bool unsq) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        int x = P.ensure_plain(in_val(op.m_input[0]));
        const Val* ax = hval(op.m_input[1]);
        need(op, ax && ax->dtype == OSG_I64 && ax->host_valid, "wrong data type of axes.");
        Shape os = V(x).shape;
        std::vector<long> axes(ax->host_i.begin(), ax->host_i.end());
        if (unsq) {
            const long orank = (long)os.size() + (long)axes.size();
            for (auto& a : axes) if (a < 0) a += orank;
            std::sort(axes.begin(), axes.end());
            for (long a : axes) { need(op, a >= 0 && a <= (long)os.size(), "invalid axis."); os.insert(os.begin() + a, 1); }
        } else {
            for (auto& a : axes) if (a < 0) a += (long)os.size();
            std::sort(axes.rbegin(), axes.rend());
            for (long a : axes) { need(op, a >= 0 && a < (long)os.size() && os[a] == 1, "invalid axis."); os.erase(os.begin() + a); }
        }
        check_out(op, os);
        P.alias(x, os, Lay::plain, op.m_output[0].m_name);
    }

    // Transpose (reference :5176-5236).  (0,2,3,1) of an NHWC tensor and (0,3,1,2) into an NHWC tensor are relabelings.
    void lower_transpose(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        auto* pm = attr(op, "perm");
        need(op, pm != nullptr, "perm attribute not found.");
        std::vector<int> perm = int_list(*pm);
        int x = in_val(op.m_input[0]);
        const Shape s = V(x).shape;
        need(op, perm.size() == s.size(), "invalid perm.");
        Shape os(s.size());
        for (size_t i = 0; i < s.size(); i++) { need(op, perm[i] >= 0 && perm[i] < (int)s.size(), "invalid index in perm."); os[i] = s[perm[i]]; }
        check_out(op, os);
        if (V(x).lay == Lay::nhwc && perm == std::vector<int>{0, 2, 3, 1}) {
            P.alias(x, os, Lay::plain, op.m_output[0].m_name);
            return;
        }
        if (V(x).lay == Lay::plain && s.size() == 4 && perm == std::vector<int>{0, 3, 1, 2}) {
            P.alias(x, os, Lay::nhwc, op.m_output[0].m_name);
            return;
        }
        x = P.ensure_plain(x);
        {   // a permutation that only moves dimensions of extent 1 leaves the memory image as it is (the head split / merge of a ONE-token decode step:
            // [1,1,H,d] <-> [1,H,1,d]): zero-copy, like a Reshape
            int prev = -1;
            bool same_image = V(x).ld == 0 && V(x).dtype != OSG_U8;
            for (size_t i = 0; same_image && i < perm.size(); i++) {
                if (s[perm[i]] == 1) continue;
                if (perm[i] < prev) same_image = false;
                prev = perm[i];
            }
            if (same_image) {
                P.alias(x, os, Lay::plain, op.m_output[0].m_name);
                return;
            }
        }
        int y = P.new_val(op.m_output[0].m_name, os, V(x).dtype, Lay::plain, V(x).batched);
        const int rank = (int)s.size() + 1;
        need(op, rank <= 6, "rank too large.");
        std::vector<long> sh(rank);
        std::vector<int> pp(rank);
        sh[0] = B(x); pp[0] = 0;
        for (int i = 1; i < rank; i++) { sh[i] = s[i - 1]; pp[i] = perm[i - 1] + 1; }
        const int es = (int)esize(V(x).dtype);
        P.add_step("Transpose " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_transpose(be.ctx, es, P.ptr(x), P.ptr(y), rank, sh.data(), pp.data()), "Transpose"); });
    }

    // Concat (reference :4140-4299)
    void lower_concat(const Operation& op) {
        need(op, !op.m_input.empty(), "wrong number of inputs.");
        auto* ax = attr(op, "axis");
        need(op, ax != nullptr, "axis attribute not found.");
        int axis = std::stoi(*ax);
        std::vector<int> xs;
        for (auto& t : op.m_input) xs.push_back(in_val(t));
        const int rank = (int)V(xs[0]).shape.size();
        if (axis < 0) axis += rank;
        need(op, axis >= 0 && axis < rank, "invalid axis.");
        bool all_nhwc = rank == 4 && axis == 1;
        for (int x : xs) all_nhwc &= V(x).lay == Lay::nhwc;
        bool batched = false;
        for (int& x : xs) { if (!all_nhwc) x = P.ensure_plain(x); batched |= V(x).batched; }
        for (int x : xs) need(op, V(x).batched == batched || N == 1, "mixing per-sample and shared operands is not implemented.");
        Shape os = V(xs[0]).shape;
        os[axis] = 0;
        for (int x : xs) {
            const Shape s = V(x).shape;
            need(op, (int)s.size() == rank, "invalid shape of inputs.");
            for (int d = 0; d < rank; d++) need(op, d == axis || s[d] == V(xs[0]).shape[d], "invalid shape of inputs.");
            os[axis] += s[axis];
        }
        int y = out_val(op, os, all_nhwc ? Lay::nhwc : Lay::plain, batched, V(xs[0]).dtype);
        const int es = (int)esize(V(y).dtype);
        // ---- operands that come straight out of convolutions are written into their slice by the convolution itself (see ConvOut) ----
        std::vector<char> placed(xs.size(), 0);
        bool any_empty = false;
        for (int x : xs) any_empty |= V(x).numel() == 0;
        if (all_nhwc && P.fusion >= 2 && indexed && !P.u8 && V(y).dtype == OSG_F16 && os[1] % 4 == 0 && P.concat_views && !any_empty) {
            long coff = 0;
            for (size_t k = 0; k < xs.size(); k++) {
                const int x = xs[k];
                const long Cx = V(x).shape[1];
                auto it = conv_producers.find(P.root_of(x));
                const bool whole = it != conv_producers.end() && it->second.y == x /* the convolution's OWN output, not a same-shape alias of it (advisor, round 3) */ && V(x).ld == 0 && V(x).view_off == 0 && V(x).numel() > 0 && V(x).batched == batched &&
                                   V(it->second.y).shape == V(x).shape && V(x).lay == Lay::nhwc && Cx % 4 == 0 && (coff * es) % 8 == 0;
                if (whole && it->second.out->dst2 < 0 && it->second.out->dst == it->second.y && it->second.out->dst_ld == 0) {
                    ConvOut& o = *it->second.out;
                    Step& st = P.steps[it->second.step];
                    const std::string& xname = op.m_input[k].m_name;
                    int others = use_count(xname) - 1;                 // readers besides this Concat
                    for (size_t k2 = 0; k2 < op.m_input.size(); k2++)
                        if (k2 != k && op.m_input[k2].m_name == xname) others = 1 << 20;   // (the same tensor twice: keep the copy path)
                    if (others == 0) {            // the Concat is its only reader: the slice is the one destination
                        o.dst = y; o.dst_ld = os[1]; o.dst_off = (size_t)coff * es;
                        for (int& wv : st.writes) if (wv == it->second.y) wv = y;
                        placed[k] = 1;
                    } else if (others < (1 << 20)) {   // other layers read the dense tensor: both destinations in one launch
                        o.dst2 = y; o.dst2_ld = os[1]; o.dst2_off = (size_t)coff * es;
                        st.writes.push_back(y);
                        placed[k] = 1;
                    }
                    if (placed[k]) {
                        st.what += " >concat";
                        viewed_steps.insert(it->second.step);
                        rs_producers.erase(P.root_of(x));
                        concat_parts[P.root_of(y)].push_back(ConcatPart{it->second.out, others == 0 ? 0 : 1, coff, it->second.step});
                    }
                }
                coff += Cx;
            }
        }
        // empty operands (the zero-length key/value caches of the LLM flow's first call, src/llm.cpp:388-402) contribute nothing
        {
            std::vector<int> live;
            for (int x : xs)
                if (V(x).numel() > 0) live.push_back(x);
            xs.swap(live);
            if (xs.empty()) return;
        }
        long outer, dst_pitch, off = 0;
        if (all_nhwc) { outer = os[2] * os[3] * (batched ? N : 1); dst_pitch = os[1]; }
        else { outer = prod(os, 0, axis) * (batched ? N : 1); dst_pitch = prod(os, axis); }
        if (std::find(placed.begin(), placed.end(), 0) != placed.end()) concat_parts.erase(P.root_of(y));   // (an operand arrives by a copy launch: no producer-side statistics)
        if (std::find(placed.begin(), placed.end(), 1) != placed.end()) {
            // (the live-operand filter above may have dropped empty operands: `placed` is indexed like the original list only when none was dropped)
            long off2 = 0;
            for (size_t k = 0; k < xs.size(); k++) {
                const int x = xs[k];
                const long inner = V(x).shape[1];
                if (!placed[k]) {
                    const long o = off2;
                    P.add_step("Concat " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_copy_2d(be.ctx, es, P.ptr(x), inner, 0, P.ptr(y), dst_pitch, o, outer, inner), "Concat"); });
                }
                off2 += inner;
            }
            return;
        }
        if (xs.size() == 2 && V(xs[0]).ld == 0 && V(xs[1]).ld == 0) {   // the skip-connection shape: both sources in one launch
            const int xa = xs[0], xb = xs[1];
            const long ia = all_nhwc ? V(xa).shape[1] : prod(V(xa).shape, axis), ib = all_nhwc ? V(xb).shape[1] : prod(V(xb).shape, axis);
            P.add_step("Concat " + op.m_name, {xa, xb}, {y}, [=, this] { be.check(be.api.osg_concat2(be.ctx, es, P.ptr(xa), ia, P.ptr(xb), ib, P.ptr(y), outer), "Concat"); });
            return;
        }
        for (int x : xs) {
            const long inner = all_nhwc ? V(x).shape[1] : prod(V(x).shape, axis);
            const long o = off;
            P.add_step("Concat " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_copy_2d(be.ctx, es, P.ptr(x), inner, 0, P.ptr(y), dst_pitch, o, outer, inner), "Concat"); });
            off += inner;
        }
    }

    // Split (reference :5999-6119): sizes from a static int64 tensor
    void lower_split(const Operation& op) {
        need(op, op.m_input.size() == 2, "wrong number of inputs.");
        int x = in_val(op.m_input[0]);
        const Val* sz = hval(op.m_input[1]);
        need(op, sz && sz->dtype == OSG_I64 && sz->host_valid && sz->host_i.size() == op.m_output.size(), "invalid split tensor.");
        int axis = 0;
        if (auto* a = attr(op, "axis")) axis = std::stoi(*a);
        const int rank = (int)V(x).shape.size();
        if (axis < 0) axis += rank;
        const bool nhwc = V(x).lay == Lay::nhwc && axis == 1;
        if (!nhwc) x = P.ensure_plain(x);
        const Shape s = V(x).shape;
        long sum = 0;
        for (auto v : sz->host_i) sum += v;
        need(op, sum == s[axis], "invalid split sizes.");
        const int es = (int)esize(V(x).dtype);
        const long outer = (nhwc ? s[2] * s[3] : prod(s, 0, axis)) * B(x), src_pitch = nhwc ? s[1] : prod(s, axis);
        const long unit = nhwc ? 1 : prod(s, axis + 1);
        long off = 0;
        for (size_t i = 0; i < op.m_output.size(); i++) {
            Shape os = s;
            os[axis] = sz->host_i[i];
            int y = out_val(op, os, nhwc ? Lay::nhwc : Lay::plain, V(x).batched, V(x).dtype, i);
            const long inner = sz->host_i[i] * unit, o = off;
            P.add_step("Split " + op.m_name, {x}, {y}, [=, this] { be.check(be.api.osg_copy_2d(be.ctx, es, P.ptr(x), src_pitch, o, P.ptr(y), inner, 0, outer, inner), "Split"); });
            off += inner;
        }
    }

    // Slice (reference :6499-6695): last or last-but-one axis, step 1, one or two axes
    void lower_slice(const Operation& op) {
        need(op, op.m_input.size() >= 3 && op.m_input.size() <= 5, "wrong number of inputs.");
        need(op, op.m_attributes.empty(), "unrecognized attribute (not implemented).");
        int x = P.ensure_plain(in_val(op.m_input[0]));
        const Val* st = hval(op.m_input[1]);
        const Val* en = hval(op.m_input[2]);
        const Val* ax = op.m_input.size() > 3 ? hval(op.m_input[3]) : nullptr;
        const Val* sp = op.m_input.size() > 4 ? hval(op.m_input[4]) : nullptr;
        need(op, st && en && st->dtype == OSG_I64 && en->dtype == OSG_I64, "wrong data type of starts.");
        const size_t na = st->host_i.size();
        need(op, (na == 1 || na == 2) && en->host_i.size() == na, "unsupported shape of starts (not implemented).");
        int cur = x;
        for (size_t k = 0; k < na; k++) {
            const Shape s = V(cur).shape;
            const int rank = (int)s.size();
            int axis = rank - 1;
            if (ax) { need(op, ax->host_i.size() == na, "unsupported shape of axes (not implemented)."); axis = (int)ax->host_i[k]; if (axis < 0) axis += rank; }
            else if (na == 2) axis = rank - 2 + (int)k;
            need(op, axis == rank - 1 || axis == rank - 2, "unsupported axes value(s): slice supported on last or last but one axis only (not implemented).");
            if (sp) need(op, sp->host_i.size() == na && sp->host_i[k] == 1, "unsupported steps value(s) (not implemented).");
            long dim = s[axis], b = st->host_i[k], e = en->host_i[k];
            if (b < 0) b += dim;
            if (b > dim - 1) b = dim - 1;
            if (e < 0) e += dim;
            if (e > dim) e = dim;
            need(op, b >= 0 && e >= 0 && b < e, "invalid value(s) in starts and/or ends.");
            Shape os = s;
            os[axis] = e - b;
            const bool last = k + 1 == na;
            int y = last ? out_val(op, os, Lay::plain, V(cur).batched, V(cur).dtype) : P.new_val("", os, V(cur).dtype, Lay::plain, V(cur).batched);
            const int es = (int)esize(V(cur).dtype);
            const long unit = prod(s, axis + 1), outer = prod(s, 0, axis) * B(cur), src_pitch = dim * unit, inner = (e - b) * unit, off = b * unit;
            const int src = cur;
            P.add_step("Slice " + op.m_name, {src}, {y}, [=, this] { be.check(be.api.osg_copy_2d(be.ctx, es, P.ptr(src), src_pitch, off, P.ptr(y), inner, 0, outer, inner), "Slice"); });
            cur = y;
        }
    }

    // Resize (reference :6120-6315): nearest / asymmetric / floor, scales or sizes
    void lower_resize(const Operation& op) {
        need(op, op.m_input.size() == 3 || op.m_input.size() == 4, "wrong number of inputs.");
        for (auto& a : op.m_attributes) {
            if (a.first == "coordinate_transformation_mode") need(op, a.second == "asymmetric", "coordinate_transformation_mode must be asymmetric (not implemented).");
            else if (a.first == "mode") need(op, a.second == "nearest", "mode must be nearest (not implemented).");
            else if (a.first == "nearest_mode") need(op, a.second == "floor", "nearest_mode must be floor (not implemented).");
            else if (a.first == "cubic_coeff_a") {}
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        int x = in_val(op.m_input[0]);
        const Shape s = V(x).shape;
        need(op, s.size() == 4 && s[0] == 1, "input must be [1,C,H,W] (not implemented).");
        Shape os = s;
        if (op.m_input.size() == 4 && !op.m_input[3].m_name.empty()) {
            const Val* sz = hval(op.m_input[3]);
            need(op, sz && sz->host_i.size() == 4, "invalid sizes.");
            for (int i = 0; i < 4; i++) os[i] = sz->host_i[i];
        } else {
            const Val* sc = hval(op.m_input[2]);
            need(op, sc && sc->host_valid && sc->host_f.size() == 4, "invalid scales.");
            need(op, sc->host_f[0] == 1.f && sc->host_f[1] == 1.f, "scales for N and C must be 1 (not implemented).");
            for (int i = 2; i < 4; i++) os[i] = (long)std::floor((float)s[i] * sc->host_f[i]);
        }
        need(op, os[0] == s[0] && os[1] == s[1], "resize of N/C not supported.");
        const bool nhwc = V(x).lay == Lay::nhwc;
        int y = out_val(op, os, V(x).lay, V(x).batched, V(x).dtype);
        const int es = (int)esize(V(x).dtype);
        const long nb = B(x);
        P.add_step("Resize " + op.m_name, {x}, {y}, [=, this] {
            be.check(be.api.osg_resize_nearest(be.ctx, es, P.ptr(x), P.ptr(y), (int)nb, (int)s[1], (int)s[2], (int)s[3], (int)os[2], (int)os[3], nhwc ? 1 : 0), "Resize");
        });
    }

    // MaxPool (reference :8075-8148 -> XnnPack::maxpool_nhwc :1536)
    void lower_maxpool(const Operation& op) {
        need(op, op.m_input.size() == 1, "wrong number of inputs.");
        std::vector<int> ks, pads = {0, 0, 0, 0}, strides = {1, 1};
        for (auto& a : op.m_attributes) {
            if (a.first == "kernel_shape") ks = int_list(a.second);
            else if (a.first == "pads") pads = int_list(a.second);
            else if (a.first == "strides") strides = int_list(a.second);
            else if (a.first == "ceil_mode") need(op, std::stoi(a.second) == 0, "ceil_mode != 0 not supported.");
            else if (a.first == "dilations") { for (int d : int_list(a.second)) need(op, d == 1, "dilations != 1 not supported."); }
            else throw std::invalid_argument(op.m_type + ": unrecognized attribute: " + a.first + ".");
        }
        need(op, ks.size() == 2 && pads.size() == 4 && strides.size() == 2, "invalid attributes.");
        int x = P.ensure_nhwc(in_val(op.m_input[0]));
        const Shape s = V(x).shape;
        const long Ho = (s[2] + pads[0] + pads[2] - ks[0]) / strides[0] + 1, Wo = (s[3] + pads[1] + pads[3] - ks[1]) / strides[1] + 1;
        int y = out_val(op, {s[0], s[1], Ho, Wo}, Lay::nhwc, V(x).batched);
        const long nb = B(x);
        P.add_step("MaxPool " + op.m_name, {x}, {y}, [=, this] {
            be.check(be.api.osg_maxpool_nhwc(be.ctx, OSG_F16, P.ptr(x), P.ptr(y), (int)nb, (int)s[2], (int)s[3], (int)s[1], ks[0], ks[1], strides[0], strides[1],
                                             pads[0], pads[1], pads[2], pads[3]),
                     "MaxPool");
        });
    }
};

// ======================================================================================================================
Plan::~Plan() {
    // A plan is normally destroyed with its last pass long waited for -- Model::run tears the plan it replaced down while the device works on the NEW plan's pass
    // (Plan::execute's hook), so neither a blanket device sync nor an early hipFree (which waits for the device) belongs at the top: host-side state first,
    // buffers back to the pools (no driver call), driver calls that may wait last.  A plan whose own pass may still be running (an exception between enqueue and
    // wait) does sync first.
    if (in_flight) be.api.osg_sync(be.ctx);
    steps.clear();
    steps.shrink_to_fit();
    delete lowering;
    lowering = nullptr;
    for (auto& o : outputs)
        if (o.dev) pool.give_class(be, o.dev, ConstPool::size_class(o.dev_bytes));      // (a pass that never ran, or threw: the buffer was not handed to a Tensor)
    for (auto& r : recyclable) pool.give(be, r.first, r.second);   // (this plan's passes are over: see above)
    for (auto& kv : registered) be.api.osg_host_unregister(be.ctx, (void*)kv.first);
    if (graph) be.api.osg_graph_destroy(graph);
    if (samp_x) be.api.osg_free(be.ctx, samp_x);
    if (samp_noise) be.api.osg_free(be.ctx, samp_noise);
    if (ring) be.free(ring);
    for (void* p : owned) be.free(p);
    if (arena && !arena_pooled) be.free(arena);
}

void Plan::build() {
    if (!fp16 && !u8)
        throw std::runtime_error("Model::run: the HIP backend implements f16 arithmetic (m_use_fp16_arithmetic, the configuration the reference uses for "
                                 "the SD UNet, src/sd.cpp:1633) and uint8 arithmetic (m_use_uint8_arithmetic, src/sd.cpp:1218); fp32 arithmetic is not implemented.");
    if (m.m_use_uint8_qdq)
        throw std::runtime_error("Model::run: m_use_uint8_qdq (uint8 storage between fp32 ops) is not implemented on the HIP backend.");
    if (m.m_requires_upcast && u8)
        throw std::runtime_error("Model::run: m_requires_upcast with uint8 arithmetic is not implemented on the HIP backend.");
    if (calibrate) {
        if (u8) throw std::invalid_argument("Model::run: m_range_data_calibrate runs in floating-point arithmetic (src/sd.cpp:1216-1222), not with m_use_uint8_arithmetic.");
        fusion = 0;      // one output per graph op, at the reference's rounding points
    }
    if (calibrate && stream_weights)   // (the streamed pass would win in execute() and measure nothing: m_range_data would stay empty without an error)
        throw std::invalid_argument("Model::run: m_range_data_calibrate cannot be combined with streamed weights (hip_stream_weights / a VRAM budget) on the HIP backend: calibrate with resident weights.");
    if (u8) {
        if (N != 1) throw std::invalid_argument("Model::run: uint8 arithmetic runs one sample per pass on the HIP backend (every pushed sample is quantised with its own scale).");
        fusion = 0;      // every op re-quantises to its own (scale, zero point): fusing ops would change codes
    }
    be.check(be.api.osg_set_autotune(be.ctx, m.m_hip_autotune ? 1 : 0), "osg_set_autotune");
    // OSG_PLAN_TIMING=1: host milliseconds of the plan-building phases on stderr (the LLM flow re-plans on every call)
    static const bool timing = std::getenv("OSG_PLAN_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    const auto t_begin = now();
    vals.reserve(m.m_ops.size() * 12 + 1024);  // belt and braces: lowering code copies shapes, never holds Val& across new_val
    lowering = new Lowering(*this);
    Lowering& L = *lowering;
    // models that re-plan on every call (m_support_dynamic_shapes: the LLM flow) reuse two things across plans, both kept in the Model's pool and both
    // functions of the graph, its constants and the options named in `key` only: the op list AFTER the fusion passes, and the constant vals that
    // load_weights() makes of the pool's resident weights (`snap_vals`: ~160 vals of a 22-layer Llama, ~1 ms to rebuild, ~0.1 ms to copy)
    std::string key;
    const bool cacheable = m.m_support_dynamic_shapes && !stream_weights && !budgeted;
    if (cacheable) {
        key = std::to_string(fusion) + "|" + std::to_string(m.m_hip_fusion_level) + (u8 ? "|u8" : "|f") + (fp16 ? "h" : "-") + (w8_resident ? "8" : "-") + (fuse_attn ? "a" : "-") +
              (sdp_attn ? "s" : "-") + (fuse_ln_gemm ? "l" : "-") + "|" + std::to_string(m.m_attention_fused_ops_parts) + "|" + std::to_string(m.m_ops.size()) + "|";
        for (auto& e : extra_outputs) key += e + ",";
        key += "|";
        if (m.m_requires_upcast)
            for (auto& op : m.m_ops) key += m.m_requires_upcast(op.m_type, op.m_name) ? '1' : '0';
    }
    const bool reuse = cacheable && pool.complete && pool.fused_valid && pool.fused_key == key && pool.snap_vals.size() == pool.fused_const_vals && !pool.snap_vals.empty();
    if (reuse) {
        vals.assign(pool.snap_vals.begin(), pool.snap_vals.end());
        weight_bytes = pool.snap_weight_bytes;
        ops = pool.fused_ops;
        L.dead.assign(ops.size(), 0);
    } else {
        ops = m.m_ops;
        L.load_weights();
    }
    const double t_weights = ms_since(t_begin);
    const size_t n_const_vals_after_load = vals.size();     // (weights are the first vals of every plan of a Model, in model order)

    // ---- graph inputs: every activation name that is consumed but never produced --------------------------------
    {
        // the names first -- a function of the graph alone, kept in the pool for the models that re-plan on every call
        std::vector<std::string> found;
        if (!(reuse && !pool.in_names.empty())) {
            std::unordered_set<std::string_view> produced, seen;   // (views into the Model's own op list: the same activation names whatever `ops` holds by now)
            produced.reserve(m.m_ops.size() * 2);
            for (auto& op : m.m_ops)
                for (auto& o : op.m_output) produced.insert(o.m_name);
            for (auto& op : m.m_ops)
                for (auto& in : op.m_input) {
                    if (in.m_name.empty() || in.m_type != TensorDataType::none || produced.count(in.m_name) || seen.count(in.m_name)) continue;
                    seen.insert(in.m_name);
                    found.push_back(in.m_name);
                }
            if (cacheable) pool.in_names = found;
        }
        for (const std::string& iname : (reuse && !pool.in_names.empty()) ? pool.in_names : found) {
            Tensor* src = nullptr;
            for (auto& t : m.m_data)
                if (t.m_name == iname) { src = &t; break; }
            if (!src) throw std::invalid_argument("Model::get_tensor_data: input tensor not found: " + iname);
            if (src->m_type != TensorDataType::float32 && src->m_type != TensorDataType::int64 && !(src->m_type == TensorDataType::float16 && fp16 && !u8))
                throw std::invalid_argument("Model::run: graph inputs must be float32, float16 (fp16 arithmetic) or int64 host tensors on the HIP backend (" + iname + ").");
            Shape shape = to_shape(src->m_shape);
            In inp;
            inp.name = iname;
            inp.host_type = src->m_type;
            inp.shape = src->m_shape;
            if (src->m_type == TensorDataType::int64) {
                // token ids / positions / masks of the LLM graphs: values known now, consumed by plan-time evaluation (Lowering::try_host_eval)
                if (u8 || N != 1) throw std::invalid_argument("Model::run: int64 graph inputs need one sample per pass and floating-point arithmetic (" + iname + ").");
                auto& iv = src->get_vector<int64_t>();
                inp.ivals.assign(iv.begin(), iv.end());
                inp.staging = inp.val = new_val(iname, shape, OSG_I64, Lay::plain, false);
                vals[inp.val].is_const = true;
                vals[inp.val].host_valid = vals[inp.val].host_only = true;
                vals[inp.val].host_i = inp.ivals;
                inputs.push_back(std::move(inp));
                continue;
            }
            if (src->m_type == TensorDataType::float16 && prod(shape) != 0 && src->m_hip_resident) {
                // an output of an earlier call that never left the device (m_hip_resident_outputs): read where it lies, no staging, no upload
                if (N != 1) throw std::invalid_argument("Model::run: float16 graph inputs need one sample per pass (" + iname + ").");
                if (src->m_hip_resident_bytes != (size_t)prod(shape) * 2) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
                inp.staging = inp.val = new_val(iname, shape, OSG_F16, Lay::plain, true);
                vals[inp.val].dptr = src->m_hip_resident.get();
                vals[inp.val].pinned = true;
                inp.resident = src->m_hip_resident;
                inputs.push_back(std::move(inp));
                continue;
            }
            if (src->m_type == TensorDataType::float16 && prod(shape) != 0) {
                // an output the caller kept in fp16 (m_outputs_convert_set excludes it) and feeds back under another name: the LLM app's
                // opkv* -> pkv* renaming (src/llm.cpp:403-407).  Uploaded as it is, no rounding step.
                if (N != 1) throw std::invalid_argument("Model::run: float16 graph inputs need one sample per pass (" + iname + ").");
                inp.staging = inp.val = new_val(iname, shape, OSG_F16, Lay::plain, true);
                vals[inp.val].dptr = small_alloc(val_bytes(inp.val));
                vals[inp.val].pinned = true;
                inputs.push_back(std::move(inp));
                continue;
            }
            if (prod(shape) == 0) {
                // an empty tensor (the first call of the LLM flow pushes zero-length key/value caches, src/llm.cpp:388-402): a val without storage
                inp.staging = inp.val = new_val(iname, shape, u8 ? OSG_U8 : OSG_F16, Lay::plain, true);
                vals[inp.val].pinned = true;
                inputs.push_back(std::move(inp));
                continue;
            }
            if (u8) {
                // a pushed fp32 input is quantised with the 0.1 % percentiles of ITS OWN data (push_tensor -> Model::quantize, reference
                // :3024-3028, :3247): done on the host in execute(), the codes are uploaded, scale / zero point live in the val
                inp.staging = inp.val = new_val(iname, shape, OSG_U8, Lay::plain, true);
                vals[inp.val].dptr = small_alloc(val_bytes(inp.val));
                vals[inp.val].pinned = true;
                vals[inp.val].qdyn = true;
                inputs.push_back(std::move(inp));
                continue;
            }
            inp.staging = new_val("", shape, OSG_F32, Lay::plain, true);
            vals[inp.staging].dptr = small_alloc(val_bytes(inp.staging));
            vals[inp.staging].pinned = true;
            // fp32 inputs are rounded to f16 when pushed with fp16 arithmetic on (reference push_tensor :3029-3034)
            inp.val = new_val(iname, shape, OSG_F16, Lay::plain, true);
            const int s = inp.staging, d = inp.val;
            const long n = total_elems(d);
            add_step("input " + iname, {s}, {d}, [this, s, d, n] {
                be.check(be.api.osg_convert(be.ctx, OSG_F32, OSG_F16, ptr(s), ptr(d), n, 1.f, 0), "osg_convert");
            });
            inputs.push_back(std::move(inp));
        }
    }

    const double ms_inputs = ms_since(t_begin) - t_weights;
    const auto t_fuse = now();
    if (!reuse) {
        if (cacheable) pool.snap_vals.assign(vals.begin(), vals.begin() + n_const_vals_after_load);   // (before the lowering hangs re-laid-out twins on them)
        L.run_fusions();
        if (cacheable) {
            pool.fused_ops = ops;
            pool.fused_key = key;
            pool.fused_const_vals = n_const_vals_after_load;
            pool.snap_weight_bytes = weight_bytes;
            pool.fused_valid = true;
        }
    }
    const double ms_fuse = ms_since(t_fuse);
    const auto t_lower = now();
    L.lower_all();
    const double ms_lower = ms_since(t_lower);

    // ---- graph outputs: produced but never consumed, plus the caller's extra outputs -> fp32, logical layout --------
    {
        std::unordered_set<std::string_view> consumed;
        consumed.reserve(ops.size() * 3);
        for (auto& op : ops)
            for (auto& in : op.m_input)
                if (!in.m_name.empty() && in.m_type == TensorDataType::none) consumed.insert(in.m_name);
        std::vector<std::string> names;
        for (auto& op : ops)
            for (auto& o : op.m_output)
                if (!consumed.count(o.m_name)) names.push_back(o.m_name);
        for (auto& e : extra_outputs)
            if (by_name.count(e) && std::find(names.begin(), names.end(), e) == names.end()) names.push_back(e);
        for (auto& nme : names) {
            int v = ensure_plain(by_name.at(nme));
            Out o;
            o.name = nme;
            o.val = v;
            for (long d : vals[v].shape) o.shape.push_back((size_t)d);
            if (!outputs_convert_set.empty() && !outputs_convert_set.count(nme) && vals[v].dtype == OSG_F16) {
                // m_outputs_convert_set (reference :8234): only the listed outputs are converted back to fp32 at the end of run(); the others
                // stay in the arithmetic type (f16 bits; here always in the logical layout).  A pinned f16 copy is what the caller reads.
                o.f32val = new_val("", vals[v].shape, OSG_F16, Lay::plain, vals[v].batched);
                if (resident_outputs && N == 1 && val_bytes(v) > 0) {   // stays on the device: its own buffer, handed to the Tensor after the pass
                    o.dev_bytes = val_bytes(v);
                    o.dev = pool.take_class(be, ConstPool::size_class(o.dev_bytes));
                    vals[o.f32val].dptr = o.dev;
                } else
                    vals[o.f32val].dptr = small_alloc(val_bytes(o.f32val));
                vals[o.f32val].pinned = true;
                o.raw16 = true;
                const int s0 = v, d0 = o.f32val;
                const size_t nbytes = val_bytes(v);
                add_step("output " + nme, {s0}, {d0}, [this, s0, d0, nbytes] { be.check(be.api.osg_copy(be.ctx, ptr(d0), ptr(s0), nbytes), "osg_copy"); });
                outputs.push_back(std::move(o));
                continue;
            }
            o.f32val = new_val("", vals[v].shape, OSG_F32, Lay::plain, vals[v].batched);
            vals[o.f32val].dptr = small_alloc(val_bytes(o.f32val));
            vals[o.f32val].pinned = true;
            const int s = v, d = o.f32val;
            const long n = total_elems(v);
            const osg_dtype sd = vals[v].dtype;
            // uint8 outputs are dequantised with the parameters their val carries at RUN time: (float)((int)q - zp) * scale (reference :8238 -> dequantize :3353)
            add_step("output " + nme, {s}, {d}, [this, s, d, n, sd] {
                const Val& q = qv(s);
                be.check(be.api.osg_convert(be.ctx, sd, OSG_F32, ptr(s), ptr(d), n, sd == OSG_U8 ? q.qscale : 1.f, sd == OSG_U8 ? q.qzp : 0), "osg_convert");
            });
            outputs.push_back(std::move(o));
        }
    }

    if (u8) {
        // the last step that reads a value whose (scale, zero point) are only known per run; the output dequantisation of a pass-through of
        // such a value counts too.  Also every step that is not safe inside a capture: none besides these (tables are built at plan time).
        for (size_t si = 0; si < steps.size(); si++)
            for (int v : steps[si].reads)
                if (v >= 0 && qv(v).qdyn) dyn_end = si + 1;
    }
    const double ms_outputs = ms_since(t_lower) - ms_lower;
    const auto t_pack = now();
    // ---- liveness + arena packing -----------------------------------------------------------------------------------
    for (size_t si = 0; si < steps.size(); si++)
        for (auto* lst : {&steps[si].reads, &steps[si].writes})
            for (int v : *lst) {
                Val& r = vals[root_of(v)];
                r.first = std::min(r.first, (int)si);
                r.last = std::max(r.last, (int)si);
            }
    struct Block { size_t off, size; };
    std::vector<Block> free_list;
    size_t top = 0;
    auto take = [&](size_t size) {
        for (size_t i = 0; i < free_list.size(); i++)
            if (free_list[i].size >= size) {
                size_t off = free_list[i].off;
                free_list[i].off += size;
                free_list[i].size -= size;
                if (!free_list[i].size) free_list.erase(free_list.begin() + i);
                return off;
            }
        if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {  // grow the trailing hole
            size_t off = free_list.back().off;
            top = off + size;
            free_list.pop_back();
            return off;
        }
        size_t off = top;
        top += size;
        return off;
    };
    auto give = [&](size_t off, size_t size) {
        free_list.push_back({off, size});
        std::sort(free_list.begin(), free_list.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < free_list.size();)
            if (free_list[i].off + free_list[i].size == free_list[i + 1].off) {
                free_list[i].size += free_list[i + 1].size;
                free_list.erase(free_list.begin() + i + 1);
            } else i++;
    };
    std::vector<std::vector<int>> born(steps.size()), dies(steps.size());
    for (size_t v = 0; v < vals.size(); v++) {
        Val& r = vals[v];
        if (r.root >= 0 || r.dptr || r.is_const || r.last < 0) continue;
        born[r.first].push_back((int)v);
        dies[r.last].push_back((int)v);
    }
    auto aligned = [](size_t b) { return (b + 255) & ~(size_t)255; };
    for (size_t si = 0; si < steps.size(); si++) {
        for (int v : born[si]) vals[v].offset = take(aligned(val_bytes(v)));
        for (int v : dies[si]) give(vals[v].offset, aligned(val_bytes(v)));
    }
    if (stream_weights) {
        // Providers serve strictly in MODEL order, launches run in PLAN order (fusions move an op's weights to where the fused launch sits):
        // before step si everything up to the last-in-model-order weight it reads has to be pulled (and sent).  flush_upto[si] = that
        // prefix of `recipes`, made monotonic.
        std::map<int, size_t> recipe_of;
        for (size_t ri = 0; ri < recipes.size(); ri++)
            if (recipes[ri].val >= 0) recipe_of[recipes[ri].val] = ri;
        flush_upto.assign(steps.size(), 0);
        size_t upto = 0;
        for (size_t si = 0; si < steps.size(); si++) {
            for (int v : steps[si].reads) {
                auto it = recipe_of.find(root_of(v));
                if (it != recipe_of.end()) upto = std::max(upto, it->second + 1);
            }
            flush_upto[si] = upto;
        }
    }
    if (budgeted && ring_weight_bytes) {
        // the streaming ring: FIFO of the over-budget weights in the order the steps read them.  It must hold what ONE step reads plus what
        // the next one is being sent (upload(i+1) overlaps compute(i)): twice the largest per-step demand, or a quarter of the streamed
        // weights up to 32 MiB if that is more
        std::vector<size_t> per_step(steps.size() + 1, 0);
        size_t worst = 0;
        for (size_t ri = 0; ri < recipes.size(); ri++) {
            const WRecipe& r = recipes[ri];
            if (!(r.ring && r.val >= 0 && vals[r.val].last >= 0)) continue;
            int sent = vals[r.val].first;        // it occupies the ring from the step whose flush sends it ...
            for (int si = 0; si < (int)steps.size(); si++)
                if (flush_upto[si] > ri) { sent = std::min(sent, si); break; }
            for (int si = sent; si <= vals[r.val].last; si++) {
                per_step[si] += ((size_t)r.count * esize(r.want) + 255) & ~(size_t)255;
                worst = std::max(worst, per_step[si]);
            }
        }
        ring_bytes = std::max<size_t>(2 * worst + 4096, std::min<size_t>((size_t)32 << 20, ring_weight_bytes / 4));
        ring = be.malloc(ring_bytes);
    }
    arena_bytes = top ? top : 256;
    if (gn_stats_bytes) { gn_stats = (char*)be.malloc(gn_stats_bytes); owned.push_back(gn_stats); }   // (zeroed at the start of every pass: zero_gn_stats)
    if (recycle) {
        arena = pooled_malloc(arena_bytes);
        arena_pooled = true;
    } else
        arena = be.malloc(arena_bytes);
    be.check(be.api.osg_sync(be.ctx), "osg_sync");
    if (timing)
        fprintf(stderr, "[plan] %zu ops -> %zu steps: weights %.2f ms, fusions %.2f ms, lowering %.2f ms, liveness + packing + arena %.2f ms, whole build %.2f ms (arena %.1f MB; graph inputs %.2f ms, outputs %.2f ms)\n", ops.size(),
                steps.size(), t_weights, ms_fuse, ms_lower, ms_since(t_pack), ms_since(t_begin), arena_bytes / 1e6, ms_inputs, ms_outputs);
}

// the producers' epilogues ADD to the statistics tables of the GroupNorms that read them (lower_group_norm): a pass starts from zero
void Plan::zero_gn_stats() {
    if (gn_stats_bytes) be.check(be.api.osg_memset(be.ctx, gn_stats, 0, gn_stats_bytes), "osg_memset");
}

void Plan::run_steps(size_t begin, size_t end) {
    // measurement aid (tools/skip_probe.sh): OSG_PLAN_SKIP=<prefix>[,<prefix>...] leaves out every step whose description starts with one of
    // the prefixes -- the pass computes garbage, its captured-graph time shows what that class of launches really costs inside the chain
    static const std::vector<std::string> skip = [] {
        std::vector<std::string> v;
        if (const char* e = std::getenv("OSG_PLAN_SKIP")) {
            std::string t = e;
            size_t b0 = 0;
            while (b0 <= t.size()) {
                size_t e0 = t.find(',', b0);
                if (e0 == std::string::npos) e0 = t.size();
                if (e0 > b0) v.push_back(t.substr(b0, e0 - b0));
                b0 = e0 + 1;
            }
        }
        return v;
    }();
    end = std::min(end, steps.size());
    if (begin == 0) zero_gn_stats();
    for (size_t si_ = begin; si_ < end; si_++) {
        Step& s = steps[si_];
        if (!skip.empty()) {
            bool sk = false;
            for (auto& pre : skip) sk |= s.what.rfind(pre, 0) == 0;
            if (sk) continue;
        }
        static const bool roctx_on = std::getenv("OSG_ROCTX") != nullptr;
        struct Range {
            HipBackend& b; bool on;
            Range(HipBackend& b_, bool on_, const char* n) : b(b_), on(on_) { if (on) b.api.osg_range_push(n); }
            ~Range() { if (on) b.api.osg_range_pop(); }
        } range(be, roctx_on, s.what.c_str());
        s.run();
        // measurement aid (round 6): OSG_PROBE_EXTRA_TRIVIAL=<k> puts k trivial launches (a 4-element convert on a private buffer) behind EVERY
        // step of the pass -- eager, captured and replayed alike: (replayed pass time with k) - (without) over k x steps = what a trivial node costs INSIDE this pass
        static const int extra_trivial = std::getenv("OSG_PROBE_EXTRA_TRIVIAL") ? atoi(std::getenv("OSG_PROBE_EXTRA_TRIVIAL")) : 0;
        if (extra_trivial > 0) {
            static void* probe = nullptr;          // (first used in the eager first pass: never allocated inside a capture; a process-lifetime 4 KiB)
            if (!probe) be.check(be.api.osg_malloc(be.ctx, 4096, &probe), "osg_malloc");
            for (int k = 0; k < extra_trivial; k++)
                be.check(be.api.osg_convert(be.ctx, OSG_F32, OSG_F16, probe, (char*)probe + 2048, 4, 1.0f, 0), "osg_convert");
        }
        // OSG_PLAN_TRACE=1 (debugging aid, eager passes only): name every step on stderr and wait for it -- a device fault then points at its launch
        static const bool trace = std::getenv("OSG_PLAN_TRACE") != nullptr;
        if (trace && !in_capture) {
            fprintf(stderr, "[step] %s\n", s.what.c_str());
            be.check(be.api.osg_sync(be.ctx), "osg_sync");
        }
    }
}

void Plan::execute(const std::function<void()>& while_device_runs) {
    // ---- m_hip_resident_outputs: the buffers of the previous execute() of THIS plan now belong to the Tensors it published (or to copies the caller
    // kept): a plan that runs again writes into buffers of its own.  (advisor, round 2: the second execute() of a compatible plan overwrote the
    // buffer the first call's Tensor owned.)  The launch closures read ptr() at run time; a captured graph has the old addresses baked in and is dropped.
    {
        bool moved = false;
        for (auto& o : outputs)
            if (o.dev_bytes && !o.dev) {
                o.dev = pool.take_class(be, ConstPool::size_class(o.dev_bytes));
                vals[o.f32val].dptr = o.dev;
                moved = true;
            }
        if (moved && graph) {
            be.api.osg_graph_destroy(graph);
            graph = nullptr;
        }
    }
    static const bool exec_times = getenv("OSG_EXEC_TIMES") != nullptr;    // developer probe: host milliseconds of the phases of this call, to stderr
    const auto t_exec = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    double ms_stage = 0, ms_enqueue = 0, ms_wait = 0;
    // ---- stage the inputs (host fp32, N samples stacked) -------------------------------------------------------------
    // Many small fp16 inputs (the LLM flow feeds 2 x layers key/value caches back every call) go up in ONE transfer when their device buffers are
    // neighbours in a small-allocation slab (they are carved out one after the other): gathered into a host block with the device's own spacing.
    char* up_lo = nullptr;
    char* up_hi = nullptr;
    {
        size_t sum = 0, cnt = 0;
        for (auto& in : inputs) {
            if (in.host_type != TensorDataType::float16 || vals[in.staging].numel() == 0 || in.resident) continue;
            char* p = (char*)ptr(in.val);
            const size_t nb = val_bytes(in.val);
            if (!up_lo || p < up_lo) up_lo = p;
            if (!up_hi || p + nb > up_hi) up_hi = p + nb;
            sum += (nb + 255) & ~(size_t)255;   // (what small_alloc hands out)
            cnt++;
        }
        // only a gap-free run inside ONE slab qualifies: nothing else may live in the range that is overwritten
        if (cnt < 4 || (size_t)(up_hi - up_lo) > sum || !in_one_slab(up_lo, up_hi)) up_lo = up_hi = nullptr;
        else io_block.assign((size_t)(up_hi - up_lo), 0);
    }
    for (auto& in : inputs) {
        Tensor* src = nullptr;
        for (auto& t : m.m_data)
            if (t.m_name == in.name) { src = &t; break; }
        if (!src) throw std::invalid_argument("Model::get_tensor_data: input tensor not found: " + in.name);
        if (src->m_type != in.host_type || src->m_shape != in.shape)
            throw std::invalid_argument("Model::run: input '" + in.name + "' changed type or shape since the plan was built.");
        if (in.host_type == TensorDataType::int64 || vals[in.staging].numel() == 0) continue;   // plan-time value / empty tensor: nothing to stage
        if (in.resident) {   // on the device already; the tensor must still be the one the plan was built on
            if (src->m_hip_resident != in.resident) throw std::invalid_argument("Model::run: input '" + in.name + "' changed since the plan was built.");
            continue;
        }
        if (in.host_type == TensorDataType::float16) {
            if (src->m_hip_resident) throw std::invalid_argument("Model::run: input '" + in.name + "' changed since the plan was built.");
            auto& vec = src->get_vector<uint16_t>();
            if (vec.size() != (size_t)vals[in.val].numel()) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
            if (up_lo) std::memcpy(io_block.data() + ((char*)ptr(in.val) - up_lo), vec.data(), vec.size() * 2);
            else be.check(be.api.osg_upload(be.ctx, ptr(in.val), vec.data(), vec.size() * 2), "osg_upload");
            continue;
        }
        const size_t per = vals[in.staging].numel() * sizeof(float);
        auto upload = [&](Tensor& t, long idx) {
            auto& vec = t.get_vector<float>();
            if (vec.size() * sizeof(float) != per) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
            if (u8) {
                std::vector<uint8_t> codes(vec.size());
                qu8::QParams qp;
                if (!qu8::quantize_dynamic(vec.data(), vec.size(), m.m_threads, codes.data(), &qp))
                    throw std::invalid_argument("Model::quantize: unable to compute the percentiles of input '" + in.name + "'.");
                vals[in.val].qscale = qp.scale;
                vals[in.val].qzp = (int)qp.zero_point;
                be.check(be.api.osg_upload(be.ctx, ptr(in.val), codes.data(), codes.size()), "osg_upload");
                return;
            }
            be.check(be.api.osg_upload(be.ctx, (char*)ptr(in.staging) + idx * per, vec.data(), per), "osg_upload");
        };
        upload(*src, 0);
        const long extra = src->m_batch ? (long)src->m_batch->size() : 0;
        if (extra + 1 != N) throw std::invalid_argument("Model::run: inconsistent m_batch.size() across two or more tensors.");
        for (long i = 0; i < extra; i++) upload((*src->m_batch)[i], i + 1);
    }
    gathered_up = up_lo ? io_block.size() : 0;
    if (up_lo) be.check(be.api.osg_upload(be.ctx, up_lo, io_block.data(), io_block.size()), "osg_upload");
    ms_stage = ms_since(t_exec);
    in_flight = true;
    // ---- run the pass -------------------------------------------------------------------------------------------------
    const bool stream_pass = stream_weights && (runs >= 1 || budgeted);
    if (m.m_ops_times_printf && stream_pass)
        throw std::invalid_argument("Model::run: m_ops_times_printf is not available with streamed weights (hip_stream_weights / a VRAM budget) on the HIP backend.");
    const bool times = m.m_ops_times_printf && !calibrate && !stream_pass;
    if (!times) be.check(be.api.osg_timer_start(be.ctx), "osg_timer_start");
    const bool print = m.m_ops_printf;
    float times_total = 0;
    if (times) {
        // m_ops_times_printf (reference :3810-3815, :8199-8214: wall milliseconds per op TYPE, printed after the last op of the pass): here the
        // DEVICE milliseconds (HIP events on the compute stream) of the launches each graph op type was lowered to, eager pass, same line format
        std::map<std::string, double> per_type;
        int idx = 0;
        zero_gn_stats();
        for (auto& s : steps) {
            if (print) printf("#%i) %s\n", idx++, s.what.c_str());
            be.api.osg_range_push(s.what.c_str());
            be.check(be.api.osg_timer_start(be.ctx), "osg_timer_start");
            s.run();
            float ms1 = 0;
            be.check(be.api.osg_timer_stop(be.ctx, &ms1), "osg_timer_stop");
            be.api.osg_range_pop();
            std::string ty = s.what.substr(0, s.what.find(' '));
            const size_t plus = ty.find('+');
            if (plus != std::string::npos) ty = ty.substr(0, plus);
            per_type[ty] += ms1;
            times_total += ms1;
        }
        printf("\033[7m > \033[0m");
        for (auto& e : per_type) printf(" %s:%f,", e.first.c_str(), e.second);
        printf("\n");
        fflush(stdout);
    } else
    if (stream_pass) {
        if (budgeted && runs == 0) m.get_wp()->on_restart();   // the plan build pulled the whole sequence once; the first pass pulls it again
        // ---- streamed-weights pass: every weight is pulled from the provider again, in model order, and sent H2D on the COPY stream
        // (pinned double-buffered staging, or straight out of a RAM provider's page-locked buffer) while the compute stream works on
        // the previous steps; a step is launched right after the uploads of ITS weights were enqueued (the compute stream waits on
        // their events), so upload(i+1) overlaps compute(i).  No hipGraph: host-side copies interleave with the launches.
        streamed_bytes = 0;
        size_t ri = 0;
        ring_occ.clear();
        ring_head = 0;
        for (size_t si = 0; si < steps.size(); si++) {
            cur_step = (int)si;
            while (ri < flush_upto[si]) restream(recipes[ri++]);
            be.check(be.api.osg_copy_fence(be.ctx), "osg_copy_fence");
            steps[si].run();
            // ring occupants whose last reader has just been enqueued: mark the compute stream here, the slot may be overwritten after it
            for (auto& o : ring_occ)
                if (o.marker < 0 && o.last <= (int)si) {
                    o.marker = next_marker;
                    next_marker = (next_marker + 1) % 256;
                    be.check(be.api.osg_marker_record(be.ctx, o.marker), "osg_marker_record");
                }
        }
        cur_step = (int)steps.size();
        while (ri < recipes.size()) restream(recipes[ri++]);   // keep the provider's sequence complete
    } else
    if (graph && !print && !calibrate) {
        if (dyn_end) run_steps(0, dyn_end);
        be.check(be.api.osg_graph_launch(be.ctx, graph), "osg_graph_launch");
    } else if (runs >= 1 && m.m_hip_use_graph && !print && !graph && !calibrate) {
        // (uint8 plans: the steps that read per-run quantisation parameters run eagerly first, see dyn_end; the rest is captured)
        if (dyn_end) run_steps(0, dyn_end);
        be.check(be.api.osg_graph_begin(be.ctx), "osg_graph_begin");
        in_capture = true;
        try {
            run_steps(dyn_end);
            in_capture = false;
        } catch (...) {
            in_capture = false;
            osg_graph* g = nullptr;
            be.api.osg_graph_end(be.ctx, &g);
            if (g) be.api.osg_graph_destroy(g);
            throw;
        }
        be.check(be.api.osg_graph_end(be.ctx, &graph), "osg_graph_end");
        be.check(be.api.osg_graph_launch(be.ctx, graph), "osg_graph_launch");
    } else if (calibrate) {
        // Model::push_tensor's calibration hook (reference :2983-3003): the 0.1 % percentiles of every tensor an op pushes widen the range
        // recorded under the OP's name.  The tensors live on the device: each is read back right after its producer (the arena recycles
        // it later) and measured by the host restatement of get_percentiles (qu8.h), chunked by the Model's thread count like the reference's.
        size_t ci = 0;
        std::vector<uint16_t> h16;
        std::vector<float> h32;
        static const std::vector<float> half_table = [] {
            std::vector<float> t(65536);
            for (unsigned k = 0; k < 65536; k++) t[k] = half_to_float((uint16_t)k);
            return t;
        }();
        const size_t workers = std::max(1u, std::thread::hardware_concurrency());
        for (size_t si = 0; si < steps.size(); si++) {
            steps[si].run();
            for (; ci < calib.size() && calib[ci].step <= (int)si; ci++) {
                const Calib& c = calib[ci];
                const size_t n = (size_t)total_elems(c.val);
                h16.resize(n);
                h32.resize(n);
                be.check(be.api.osg_download(be.ctx, h16.data(), ptr(c.val), n * 2), "osg_download");
                for (size_t k = 0; k < n; k++) h32[k] = half_table[h16[k]];
                auto r = qu8::percentiles_fast(h32.data(), n, 0.001f, 0.001f, m.m_threads, workers);
                if (!r) continue;
                auto it = m.m_range_data.find(c.op);
                if (it == m.m_range_data.end()) m.m_range_data[c.op] = *r;
                else {
                    if (r->first < it->second.first) it->second.first = r->first;
                    if (r->second > it->second.second) it->second.second = r->second;
                }
            }
        }
    } else if (!print) {
        run_steps();
    } else {
        int idx = 0;
        zero_gn_stats();
        for (auto& s : steps) {
            printf("#%i) %s\n", idx++, s.what.c_str());
            be.api.osg_range_push(s.what.c_str());
            s.run();
            be.api.osg_range_pop();
        }
    }
    float ms = times_total;
    ms_enqueue = ms_since(t_exec);
    if (while_device_runs) while_device_runs();
    if (!times) be.check(be.api.osg_timer_stop(be.ctx, &ms), "osg_timer_stop");
    in_flight = false;
    ms_wait = ms_since(t_exec);
    m_last_ms = ms;
    runs++;
    // ---- consume the inputs, publish the outputs as fp32 host tensors in the logical (NCHW) layout (reference :8217-8263) --
    for (auto& in : inputs)
        for (size_t i = 0; i < m.m_data.size(); i++)
            if (m.m_data[i].m_name == in.name) { m.m_data.erase(m.m_data.begin() + i); break; }
    // ... the same for many small outputs (the new caches): one transfer of the slab range that holds their staging buffers, split on the host
    char* dn_lo = nullptr;
    char* dn_hi = nullptr;
    if (outputs.size() >= 4) {
        size_t sum = 0;
        for (auto& o : outputs) {
            if (o.dev) continue;
            char* p = (char*)ptr(o.f32val);
            const size_t nb = val_bytes(o.f32val);   // (all N samples)
            if (!dn_lo || p < dn_lo) dn_lo = p;
            if (!dn_hi || p + nb > dn_hi) dn_hi = p + nb;
            sum += (nb + 255) & ~(size_t)255;
        }
        if ((size_t)(dn_hi - dn_lo) > sum || !in_one_slab(dn_lo, dn_hi)) dn_lo = dn_hi = nullptr;
        else {
            io_block.resize((size_t)(dn_hi - dn_lo));
            be.check(be.api.osg_download(be.ctx, io_block.data(), dn_lo, io_block.size()), "osg_download");
        }
    }
    gathered_down = dn_lo ? io_block.size() : 0;
    auto fetch = [&](void* host, const char* dev, size_t bytes) {
        if (dn_lo) std::memcpy(host, io_block.data() + (dev - dn_lo), bytes);
        else be.check(be.api.osg_download(be.ctx, host, dev, bytes), "osg_download");
    };
    for (auto& o : outputs) {
        if (o.dev) {
            // m_hip_resident_outputs: the Tensor in m_data owns the device buffer from here on (freed with the last copy of the Tensor, as long as the
            // Model lives); its host vector stays empty
            Tensor t;
            t.m_name = o.name;
            t.m_shape = o.shape;
            t.set_vector(tensor_vector<uint16_t>());
            HipBackend* bp = &be;
            ConstPool* pp = &pool;
            const size_t cls = ConstPool::size_class(o.dev_bytes);
            t.m_hip_resident = std::shared_ptr<void>(o.dev, [alive = std::weak_ptr<bool>(m.m_alive), bp, pp, cls](void* p) {
                if (alive.lock()) pp->give_class(*bp, p, cls);     // (back to the Model's pool: the call after next takes it again)
            });
            t.m_hip_resident_bytes = o.dev_bytes;
            o.dev = nullptr;
            for (size_t i = 0; i < m.m_data.size(); i++)
                if (m.m_data[i].m_name == o.name) { m.m_data.erase(m.m_data.begin() + i); break; }
            m.m_data.push_back(std::move(t));
            continue;
        }
        const size_t per_elems = (size_t)vals[o.f32val].numel();
        const long nb = vals[o.f32val].batched ? N : 1;
        Tensor first;
        for (long i = 0; i < nb; i++) {
            Tensor t;
            t.m_name = o.name;
            t.m_shape = o.shape;
            if (o.raw16) {
                tensor_vector<uint16_t> host(per_elems);
                fetch(host.data(), (char*)ptr(o.f32val) + i * per_elems * 2, per_elems * 2);
                t.set_vector(std::move(host));
            } else {
                tensor_vector<float> host(per_elems);
                fetch(host.data(), (char*)ptr(o.f32val) + i * per_elems * sizeof(float), per_elems * sizeof(float));
                t.set_vector(std::move(host));
            }
            if (i == 0) first = std::move(t);
            else {
                if (!first.m_batch) first.m_batch = std::make_shared<std::vector<Tensor>>();
                first.m_batch->push_back(std::move(t));
            }
        }
        for (size_t i = 0; i < m.m_data.size(); i++)
            if (m.m_data[i].m_name == o.name) { m.m_data.erase(m.m_data.begin() + i); break; }
        m.m_data.push_back(std::move(first));
    }
    if (exec_times)
        fprintf(stderr, "[exec] inputs staged %.3f ms, pass enqueued +%.3f, device done +%.3f (device %.3f), outputs published +%.3f\n", ms_stage, ms_enqueue - ms_stage,
                ms_wait - ms_enqueue, (double)m_last_ms, ms_since(t_exec) - ms_wait);
}

void Plan::restream(const WRecipe& r) {
    WeightsProvider* wp = m.get_wp();
    detail::dispatch_dtype(r.ty, [&](auto tag) {
        using T = typename decltype(tag)::type;
        const size_t bytes = (size_t)r.count * sizeof(T);
        auto send = [&](const T* host, bool stable) {
            if (r.val < 0 || r.resident) return;        // fetched to keep the provider's sequence; already on the device
            if (r.ring) {
                if (vals[r.val].last < 0) return;       // nobody reads it in this plan
                // FIFO slot in the ring; whatever it overlaps must have been read by launches that are at least enqueued (else the ring
                // is too small for one step's weights) and the COPY stream waits for those launches before it overwrites them
                const size_t need = (bytes + 255) & ~(size_t)255;
                if (need > ring_bytes) throw std::runtime_error("Model::run: a weight is larger than the VRAM streaming ring.");
                if (ring_head + need > ring_bytes) ring_head = 0;
                for (size_t k = 0; k < ring_occ.size();) {
                    RingOcc& o = ring_occ[k];
                    if (o.off < ring_head + need && ring_head < o.off + o.size) {
                        if (o.last >= cur_step) throw std::runtime_error("Model::run: the VRAM budget leaves no room for the weights one step reads (raise m_vram_to_use).");
                        if (o.marker >= 0) be.check(be.api.osg_copy_wait_marker(be.ctx, o.marker), "osg_copy_wait_marker");
                        ring_occ.erase(ring_occ.begin() + k);
                    } else
                        k++;
                }
                vals[r.val].dptr = (char*)ring + ring_head;
                ring_occ.push_back(RingOcc{ring_head, need, vals[r.val].last, -1});
                ring_head += need;
            }
            void* dst = r.raw ? r.raw : vals[r.val].dptr;
            if (stable) {   // provider-owned memory that outlives the pass: page-lock once, DMA without a staging copy
                auto it = registered.find(host);
                if (it == registered.end()) {
                    be.check(be.api.osg_host_register(be.ctx, (void*)host, bytes), "osg_host_register");
                    registered[host] = bytes;
                }
                be.check(be.api.osg_upload_pinned_async(be.ctx, dst, host, bytes), "osg_upload_pinned_async");   // (fenced once per step, Plan::execute)
                if (r.raw) be.check(be.api.osg_copy_fence(be.ctx), "osg_copy_fence");
            } else {
                be.check(be.api.osg_upload(be.ctx, dst, host, bytes), "osg_upload");
            }
            if (r.raw) be.check(be.api.osg_convert(be.ctx, r.have, r.want, r.raw, vals[r.val].dptr, r.count, r.scale, r.zp), "osg_convert");
            streamed_bytes += bytes;
        };
        if (wp->supports_getptr()) {
            std::shared_ptr<tensor_vector<T>> sp;
            if constexpr (std::is_same_v<T, uint8_t>) sp = wp->getptr_uint8(r.fn);
            else if constexpr (std::is_same_v<T, uint16_t>) sp = wp->getptr_float16(r.fn);
            else if constexpr (std::is_same_v<T, float>) sp = wp->getptr_float32(r.fn);
            else sp = wp->getptr_int64(r.fn);
            if (r.val >= 0 && (long)sp->size() != r.count) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
            send(sp->data(), true);
        } else {
            tensor_vector<T> data;
            if constexpr (std::is_same_v<T, uint8_t>) data = wp->get_uint8(r.fn);
            else if constexpr (std::is_same_v<T, uint16_t>) data = wp->get_float16(r.fn);
            else if constexpr (std::is_same_v<T, float>) data = wp->get_float32(r.fn);
            else data = wp->get_int64(r.fn);
            if (r.val >= 0 && (long)data.size() != r.count) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
            send(data.data(), false);   // osg_upload copies into pinned staging before returning: `data` may die here
        }
    });
}

// in_flight from the first enqueue to the wait at the end of the scope; an exception on the way leaves it set (the destructor then waits for the device)
namespace {
struct FlightGuard {
    bool& f;
    explicit FlightGuard(bool& f_) : f(f_) { f = true; }
    ~FlightGuard() { if (!std::uncaught_exceptions()) f = false; }
};
}  // namespace

void Plan::replay(int n, float* ms_each) {
    if (!graph) throw std::runtime_error("Model::hip_replay: no captured pass yet (run() at least twice with hip_use_graph on).");
    FlightGuard flight(in_flight);
    if (!ms_each) {  // back-to-back launches, one event pair around all of them
        be.check(be.api.osg_timer_start(be.ctx), "osg_timer_start");
        for (int i = 0; i < n; i++) be.check(be.api.osg_graph_launch(be.ctx, graph), "osg_graph_launch");
        float ms = 0;
        be.check(be.api.osg_timer_stop(be.ctx, &ms), "osg_timer_stop");
        m_last_ms = n > 0 ? ms / n : 0;
        return;
    }
    for (int i = 0; i < n; i++) {
        be.check(be.api.osg_timer_start(be.ctx), "osg_timer_start");
        be.check(be.api.osg_graph_launch(be.ctx, graph), "osg_graph_launch");
        float ms = 0;
        be.check(be.api.osg_timer_stop(be.ctx, &ms), "osg_timer_stop");
        ms_each[i] = ms;
        m_last_ms = ms;
    }
}

void Plan::set_input(const std::string& name, long index, const float* data, size_t count) {
    if (u8) throw std::runtime_error("Model::hip_set_input: not available with uint8 arithmetic (inputs are quantised per run).");
    for (auto& in : inputs)
        if (in.name == name) {
            const size_t per = (size_t)vals[in.staging].numel();
            if (count != per) throw std::invalid_argument("Model::get_tensor_data: mismatch between tensor shape and data size.");
            if (index < 0 || index >= N) throw std::invalid_argument("Model::hip_set_input: sample index out of range.");
            be.check(be.api.osg_upload(be.ctx, (char*)ptr(in.staging) + index * per * sizeof(float), data, per * sizeof(float)), "osg_upload");
            return;
        }
    throw std::invalid_argument("Model::get_tensor_data: input tensor not found: " + name);
}

double Plan::sampler_loop(const std::string& sample_name, const std::string& timestep_name, const std::string& out_name, int n_steps, int prompts,
                          float* x, const float* noise, const float* c_in, const float* c_out, const float* t, const float* sigma, const float* d_sigma,
                          const float* sigma_up, float guidance, const float* clip) {
    if (runs < 1) throw std::runtime_error("Model::hip_sampler_loop: run() once first (the context inputs must be resident).");
    if (stream_weights) throw std::runtime_error("Model::hip_sampler_loop: not available in streamed-weights mode.");
    if (u8) throw std::runtime_error("Model::hip_sampler_loop: not available with uint8 arithmetic.");
    if (prompts <= 0 || 2L * prompts != N) throw std::invalid_argument("Model::hip_sampler_loop: the plan's batch must be 2 * prompts (cond, uncond per prompt).");
    FlightGuard flight(in_flight);
    const In *in_s = nullptr, *in_t = nullptr;
    for (auto& in : inputs) {
        if (in.name == sample_name) in_s = &in;
        if (in.name == timestep_name) in_t = &in;
    }
    const Out* out = nullptr;
    for (auto& o : outputs)
        if (o.name == out_name) out = &o;
    if (!in_s || !in_t || !out) throw std::invalid_argument("Model::hip_sampler_loop: input/output tensor not found.");
    if (out->raw16) throw std::invalid_argument("Model::hip_sampler_loop: the output is excluded from the fp32 conversion (m_outputs_convert_set).");
    const long L = vals[in_s->staging].numel(), TL = vals[in_t->staging].numel();
    if (vals[out->f32val].numel() != L || !vals[out->f32val].batched)
        throw std::invalid_argument("Model::hip_sampler_loop: the output must have the shape of the sample input.");
    const size_t xb = (size_t)prompts * L * sizeof(float), nb = (size_t)n_steps * xb;
    auto grow = [&](void*& p, size_t& have, size_t need) {
        if (have >= need) return;
        if (p) be.check(be.api.osg_free(be.ctx, p), "osg_free");
        p = nullptr;
        have = 0;
        be.check(be.api.osg_malloc(be.ctx, need, &p), "osg_malloc");
        have = need;
    };
    grow(samp_x, samp_x_bytes, xb);
    if (noise) grow(samp_noise, samp_noise_bytes, nb);
    be.check(be.api.osg_upload(be.ctx, samp_x, x, xb), "osg_upload");
    if (noise) be.check(be.api.osg_upload(be.ctx, samp_noise, noise, nb), "osg_upload");
    be.check(be.api.osg_timer_start(be.ctx), "osg_timer_start");
    for (int i = 0; i < n_steps; i++) {
        be.check(be.api.osg_sampler_prepare(be.ctx, (const float*)samp_x, (float*)ptr(in_s->staging), (float*)ptr(in_t->staging), prompts, L, c_in[i], t[i], TL),
                 "osg_sampler_prepare");
        if (graph) be.check(be.api.osg_graph_launch(be.ctx, graph), "osg_graph_launch");
        else run_steps();
        const bool with_noise = noise != nullptr;
        be.check(be.api.osg_sampler_cfg_euler_a(be.ctx, (float*)samp_x, (const float*)ptr(out->f32val),
                                               with_noise ? (const float*)samp_noise + (size_t)i * prompts * L : nullptr, prompts, L, c_out[i], guidance,
                                               sigma[i], d_sigma[i], sigma_up[i], clip ? clip[i] : 0.f),
                 "osg_sampler_cfg_euler_a");
    }
    float ms = 0;
    be.check(be.api.osg_timer_stop(be.ctx, &ms), "osg_timer_stop");
    be.check(be.api.osg_download(be.ctx, x, samp_x, xb), "osg_download");
    runs += n_steps;
    m_last_ms = n_steps > 0 ? ms / n_steps : 0;
    return ms;
}

std::string Plan::info() const {
    std::string out;
    char buf[128];
    for (size_t i = 0; i < steps.size(); i++) {
        const Step& s = steps[i];
        snprintf(buf, sizeof buf, "step %zu reads=", i);
        out += buf;
        for (size_t k = 0; k < s.reads.size(); k++) out += (k ? "," : "") + std::to_string(root_of(s.reads[k]));
        out += " writes=";
        for (size_t k = 0; k < s.writes.size(); k++) out += (k ? "," : "") + std::to_string(root_of(s.writes[k]));
        out += " | " + s.what + "\n";
    }
    for (size_t v = 0; v < vals.size(); v++) {
        const Val& r = vals[v];
        if (r.root >= 0 || r.dptr || r.is_const || r.last < 0) continue;
        snprintf(buf, sizeof buf, "val %zu offset=%zu bytes=%zu first=%d last=%d\n", v, r.offset, val_bytes((int)v), r.first, r.last);
        out += buf;
    }
    out += "arena " + std::to_string(arena_bytes) + "\n";
    return out;
}

std::string Plan::profile(int reps) {
    if (runs < 1) throw std::runtime_error("Model::hip_profile: run() once first (inputs must be resident).");
    FlightGuard flight(in_flight);
    std::vector<double> acc(steps.size(), 0.0);
    // every step between two timestamps on the compute stream, the whole pass enqueued back to back (the queue stays full, as inside the
    // captured graph): a step's figure = its kernels + the dependency gap to its predecessor, no idle-launch latency from host round trips.
    // Passes of more than 4000 steps fall back to one synchronised measurement per step.
    const bool chained = steps.size() < 4000 && !stream_weights;
    for (int r = 0; r < reps; r++) {
        zero_gn_stats();
        if (chained) {
            be.check(be.api.osg_timer_mark(be.ctx, 0), "osg_timer_mark");
            for (size_t i = 0; i < steps.size(); i++) {
                steps[i].run();
                be.check(be.api.osg_timer_mark(be.ctx, (int)i + 1), "osg_timer_mark");
            }
            for (size_t i = 0; i < steps.size(); i++) {
                float ms = 0;
                be.check(be.api.osg_timer_between(be.ctx, (int)i, (int)i + 1, &ms), "osg_timer_between");
                acc[i] += ms;
            }
        } else
            for (size_t i = 0; i < steps.size(); i++) {
                be.check(be.api.osg_timer_start(be.ctx), "osg_timer_start");
                steps[i].run();
                float ms = 0;
                be.check(be.api.osg_timer_stop(be.ctx, &ms), "osg_timer_stop");
                acc[i] += ms;
            }
    }
    std::string out;
    char buf[256];
    for (size_t i = 0; i < steps.size(); i++) {
        double bytes = 0;
        for (int v : steps[i].reads) bytes += (double)val_bytes(v);
        for (int v : steps[i].writes) bytes += (double)val_bytes(v);
        snprintf(buf, sizeof buf, "%.6f\t%.0f\t%.0f\t", acc[i] / reps, steps[i].flops, bytes);
        out += buf;
        out += steps[i].what;
        out += "\n";
    }
    return out;
}

}  // namespace onnxstream
