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

#include "lowering_graph.inc"
#include "lowering_u8.inc"
#include "lowering_ops.inc"
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

}  // namespace onnxstream
