// plan.h -- the device execution plan behind Model::run().
//
// The reference interprets the graph op by op on the host, allocating every output and converting layouts/dtypes
// around each call (reference src/onnxstream.cpp:3550-8269).  On MI355X that shape of execution is launch- and
// PCIe-bound, so this backend lowers the WHOLE graph once:
//   parse (model.cpp) -> fetch weights through the WeightsProvider in model order and make them resident in HBM
//   -> graph-level fusion (SiLU, GroupNorm, LayerNorm, GEGLU, Linear+bias+residual, attention incl. head split/merge)
//   -> lowering to a list of kernel launches over device-resident f16 activations (NHWC around convolutions)
//   -> liveness-based packing of all activations into one HBM arena
//   -> eager first pass, then capture into a hipGraph that later passes replay.
#pragma once

#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <unordered_map>
#include <set>
#include <string>
#include <vector>

#include "backend.h"
#include "onnxstream.h"

namespace onnxstream {

enum class Lay { plain, nhwc };

struct Val {
    std::string name;
    std::vector<long> shape;      // logical (ONNX) shape of ONE sample
    osg_dtype dtype = OSG_F16;
    Lay lay = Lay::plain;         // nhwc: 4-D logical [n,C,H,W] stored as [n,H,W,C]
    bool batched = false;         // one copy per pushed sample, stacked contiguously
    bool is_const = false;        // resident weight / constant
    int root = -1;                // >=0: alias sharing the storage of another val
    void* dptr = nullptr;         // constants & pinned buffers: own allocation
    size_t offset = 0;            // activations: offset inside the arena
    bool pinned = false;          // graph input/output staging: never recycled
    int first = 1 << 30, last = -1;
    std::vector<float> host_f;    // small constants, available at plan time
    std::vector<int64_t> host_i;
    bool host_valid = false;
    bool up32 = false;            // fp32 result of an op the Model's m_requires_upcast flags (LLM layer norms): rounded to f16 when an unflagged op reads it
    bool host_only = false;       // value computed at plan time (Shape / Gather / Cast / Concat chains over shapes): no device storage until a launch asks for it
    int as_plain = -1, as_nhwc = -1, as_dense = -1;
    // column-slice view of a wider 2-D buffer (merged projections): rows are `ld` elements apart, starting `view_off` bytes into the root
    long ld = 0;
    size_t view_off = 0;
    float qscale = 1.f;           // dtype == OSG_U8: value = (q - qzp) * qscale  (weights: from model.txt; uint8 activations: from range_data / the
    int qzp = 0;                  //   dynamic quantisation of a pushed input)
    int qsrc = -1;                // >= 0: the val whose (qscale, qzp) this one shares (Reshape / Transpose / Resize carry them over) -- read at RUN time
    bool qdyn = false;            // parameters change from run to run (a pushed input and what merely re-arranges it)
    int as_nk_u8 = -1;            // [N,K] twin of a [K,N] uint8 matrix
    // a VIRTUAL Expand (grouped-query attention's repeat_kv: [1,Hkv,1,S,d] -> [1,Hkv,rep,S,d], read only by ScaledDotProductAttention through a Reshape):
    // never materialised -- the attention launch reads `rep_src` with Hkv heads (plan.cpp lower_expand / lower_sdpa)
    int rep_src = -1;
    long rep = 1;
    long numel() const { long n = 1; for (auto d : shape) n *= d; return n; }
};

struct Step {
    std::string what;
    std::function<void()> run;
    std::vector<int> reads, writes;
    double flops = 0;             // algorithmic 2*MAC count of a contraction step (0 for memory-bound steps)
};

struct Lowering;

// Device-resident constants that outlive a Plan (owned by the Model, next to the backend): a plan is rebuilt when the number of pushed
// samples, the input shapes or a hip_* option change, and a rebuild must not go back to the WeightsProvider -- by then a strictly
// sequential provider is exhausted and, with m_use_ops_cache, the host copies were remove()d (reference: the ops cache keeps the packed
// weights across runs, src/onnxstream.cpp:4556-4569).  `base` = weights as pulled through the provider ("file|dtype"), `derived` = the
// re-laid-out / merged / folded copies the lowering makes of them, by tag.  Not used in streamed-weights mode.
struct ConstPool {
    struct Base {
        void* dptr = nullptr;
        size_t bytes = 0;
        std::vector<float> host_f;
        std::vector<int64_t> host_i;
        bool host_valid = false;
    };
    std::map<std::string, Base> base;
    std::map<std::string, std::pair<void*, size_t>> derived;
    std::vector<TensorDataType> occ_types;   // resolved storage type of every weight occurrence, model order
    bool complete = false;                   // every occurrence of the graph went through the provider once
    size_t bytes = 0;
    // the op list AFTER the fusion passes, for models that re-plan on every call (m_support_dynamic_shapes: the LLM flow): the passes match graph
    // structure, constants and declared shapes -- none of which change between calls -- so their result is reused while `fused_key` (every option
    // they read, the extra outputs, the per-op m_requires_upcast answers) and the number of constant vals stay the same
    std::string fused_key;
    std::vector<Operation> fused_ops;
    size_t fused_const_vals = 0;
    bool fused_valid = false;
    std::vector<Val> snap_vals;              // the constant vals load_weights() builds from `base`, as they are right after it (same validity as fused_ops)
    size_t snap_weight_bytes = 0;
    std::vector<std::string> in_names;       // the graph's input names (consumed, never produced), in discovery order
    // device buffers (arena, small-allocation slabs) of a destroyed plan of such a model, handed to the next one: a hipMalloc + hipFree pair per buffer
    // and call otherwise
    std::vector<std::pair<void*, size_t>> spare;
    void* take(size_t bytes, size_t* got) {
        int best = -1;
        for (size_t i = 0; i < spare.size(); i++)
            if (spare[i].second >= bytes && (best < 0 || spare[i].second < spare[best].second)) best = (int)i;
        if (best < 0 || spare[best].second > 4 * bytes + ((size_t)8 << 20)) return nullptr;   // (no 1 GiB arena for a 1 MiB request)
        void* p = spare[best].first;
        *got = spare[best].second;
        spare.erase(spare.begin() + best);
        return p;
    }
    // ... and the buffers of device-resident outputs (Model::m_hip_resident_outputs), by size class: a cache grows by one row per call, so its buffer is
    // sized to the next power of two and the buffers of the call before last (released when their plan goes) serve the call that comes
    std::map<size_t, std::vector<void*>> spare_class;
    static size_t size_class(size_t bytes) {
        size_t c = 4096;
        while (c < bytes) c <<= 1;
        return c;
    }
    void* take_class(HipBackend& be, size_t cls) {
        auto it = spare_class.find(cls);
        if (it != spare_class.end() && !it->second.empty()) {
            void* p = it->second.back();
            it->second.pop_back();
            return p;
        }
        return be.malloc(cls);
    }
    void give_class(HipBackend& be, void* p, size_t cls) {
        auto& v = spare_class[cls];
        if (v.size() >= 512) be.free(p);
        else v.push_back(p);
    }
    void give(HipBackend& be, void* p, size_t bytes) {
        if (spare.size() >= 8) {   // keep the newest: drop the smallest
            size_t k = 0;
            for (size_t i = 1; i < spare.size(); i++)
                if (spare[i].second < spare[k].second) k = i;
            be.free(spare[k].first);
            spare.erase(spare.begin() + k);
        }
        spare.push_back({p, bytes});
    }
    void clear(HipBackend& be) {
        for (auto& kv : base) be.free(kv.second.dptr);
        for (auto& kv : derived) be.free(kv.second.first);
        for (auto& sp : spare) be.free(sp.first);
        spare.clear();
        for (auto& kv : spare_class)
            for (void* p : kv.second) be.free(p);
        spare_class.clear();
        base.clear(); derived.clear(); occ_types.clear();
        complete = false; bytes = 0;
        fused_valid = false; fused_ops.clear(); fused_key.clear(); snap_vals.clear(); snap_weight_bytes = 0; in_names.clear();
    }
};

// f16 bits -> float (host side: constants, folded weights, the sampler's table); shared by plan.cpp and plan_run.cpp
inline float half_to_float(uint16_t h) {
    uint32_t sign = (h >> 15) & 1, exp = (h >> 10) & 0x1f, man = h & 0x3ff;
    float v;
    if (exp == 0) v = std::ldexp((float)man, -24);
    else if (exp == 31) v = man ? NAN : INFINITY;
    else v = std::ldexp((float)(man | 0x400), (int)exp - 25);
    return sign ? -v : v;
}

struct Plan {
    Plan(Model& m, HipBackend& be, ConstPool& pool, size_t batch);
    ~Plan();
    void build();
    // while_device_runs: host work the caller wants done between "the pass is enqueued" and "wait for it" (Model::run destroys the plan it replaced there:
    // the LLM flow re-plans on every call, and tearing the old plan down took 0.8 ms of every call's critical path)
    void execute(const std::function<void()>& while_device_runs = nullptr);
    // relaunch the captured pass `n` times on the inputs already resident in HBM; per-launch device ms (HIP events)
    void replay(int n, float* ms_each);
    // the reference app's denoising loop with the CFG combine and the Euler-Ancestral update on the device: `steps`
    // passes enqueued back to back, one host sync at the end.  The plan's batch must be 2*prompts (push 2p = cond, 2p+1 = uncond of
    // prompt p) and every other input (context, ...) must already be resident from an earlier run().  Per-step scalars come from the
    // caller (who owns the schedule): c_in, c_out, t, sigma = sigma_i, d_sigma = sigma_down - sigma_i, sigma_up, clip (NULL or per-step clamp of the new latents, 0 = none).  x: [prompts, L] fp32 host, updated
    // in place; noise: [steps, prompts, L] fp32 host.  Returns the device time of the whole loop in ms.
    // overwrite sample `index` of a graph input in its device staging (what run() does for every pushed tensor) without running a pass
    void set_input(const std::string& name, long index, const float* data, size_t count);
    double sampler_loop(const std::string& sample_name, const std::string& timestep_name, const std::string& out_name, int n_steps, int prompts,
                        float* x, const float* noise, const float* c_in, const float* c_out, const float* t, const float* sigma, const float* d_sigma,
                        const float* sigma_up, float guidance, const float* clip);
    // eager pass with HIP events around every step, `reps` times; "ms<TAB>flops<TAB>bytes<TAB>what" per line (ms = mean)
    std::string profile(int reps);
    // plan introspection for the CPU tests of the host logic (tests/test_planner_cpu.py): one line per step
    //   "step <i> reads=<v,...> writes=<v,...> | <what>"   (val ids are ROOT vals)
    // and per arena val  "val <id> offset=<o> bytes=<b> first=<f> last=<l>", then "arena <bytes>".
    std::string info() const;
    bool compatible(Model& m, size_t batch) const;
    size_t kernel_count() const { return steps.size(); }
    double last_ms() const { return m_last_ms; }

    Model& m;
    HipBackend& be;
    ConstPool& pool;
    long N;  // batch (number of samples pushed under each input name)

    std::vector<Operation> ops;  // working copy of the graph (mutated by the fusion passes)
    std::vector<Val> vals;
    std::unordered_map<std::string, int> by_name;
    std::vector<Step> steps;
    std::vector<char> io_block;   // host side of the gathered input upload / output download of execute()
    size_t gathered_up = 0, gathered_down = 0;   // bytes the last execute() moved that way (0: buffer by buffer)
    std::vector<std::pair<char*, size_t>> slabs;   // the small-allocation slabs of this plan
    bool in_one_slab(const char* lo, const char* hi) const;
    std::vector<void*> owned;     // device allocations owned by the plan (weights, staging, outputs)
    std::vector<std::pair<void*, size_t>> recyclable;   // ... those that go back to the pool's spare list when the model re-plans on every call (slabs, arena)
    bool recycle = false, arena_pooled = false;
    void* pooled_malloc(size_t bytes);
    // small per-plan buffers (input staging, pinned outputs, index / ones / mask constants) are carved out of 8 MiB slabs: a plan of the LLM flow
    // has ~100 of them and is rebuilt on every call -- one hipMalloc each was 2/3 of the rebuild time
    void* small_alloc(size_t bytes);
    char* slab = nullptr;
    size_t slab_left = 0;
    void* arena = nullptr;
    size_t arena_bytes = 0, weight_bytes = 0;

    // streamed-weights mode: how each resident constant is (re)filled from the provider, in the provider's (model) order
    struct WRecipe {
        int val = -1;                 // -1: a repeated occurrence -- fetched (providers serve strictly in order) but not uploaded
        std::string fn;
        TensorDataType ty = TensorDataType::none;
        osg_dtype have = OSG_F16, want = OSG_F16;
        long count = 0;
        float scale = 1.f;
        int zp = 0;
        void* raw = nullptr;          // device staging for weights that need a dtype conversion after the copy
        bool resident = false;        // VRAM-budget mode: inside the budget -- uploaded once, later passes only fetch it from the provider
        bool ring = false;            // VRAM-budget mode: over the budget -- lives in the streaming ring, re-sent every pass
    };
    // ---- VRAM budget (CudaOptions::m_vram_to_use) ----
    size_t vram_budget = 0, resident_bytes = 0, ring_weight_bytes = 0;
    bool budgeted = false;
    void* ring = nullptr;
    size_t ring_bytes = 0, ring_head = 0;
    struct RingOcc { size_t off, size; int last; int marker; };
    std::vector<RingOcc> ring_occ;
    int next_marker = 0, cur_step = 0;
    std::vector<WRecipe> recipes;
    std::vector<size_t> flush_upto;             // per step: how many leading recipes must have been pulled / sent before it runs
    std::map<const void*, size_t> registered;   // provider host buffers page-locked for zero-copy DMA
    size_t streamed_bytes = 0;
    void restream(const WRecipe& r);

    // ivals: an int64 graph input (LLM graphs: input_ids, position_ids, attention_mask) is a PLAN-TIME value -- its numbers feed Gather indices and
    // mask subgraphs that are evaluated while the plan is built, so a plan is only reused for the same numbers (Plan::compatible)
    struct In {
        std::string name; int val; int staging; TensorDataType host_type; std::vector<size_t> shape; std::vector<int64_t> ivals;
        std::shared_ptr<void> resident;   // a device-resident tensor (Tensor::m_hip_resident) read where it lies; kept alive by the plan that reads it
    };
    struct Out {
        std::string name; int val; int f32val; std::vector<size_t> shape; bool raw16 = false;
        void* dev = nullptr;              // m_hip_resident_outputs: the buffer this output is left in (handed to the Tensor in m_data after the pass)
        size_t dev_bytes = 0;
    };
    bool resident_outputs = false;
    struct Calib { int step; std::string op; int val; };   // m_range_data_calibrate: val is measured after `step`, range kept under the op's name
    std::vector<Calib> calib;
    std::vector<In> inputs;
    std::vector<Out> outputs;

    void* samp_x = nullptr;       // sampler_loop state: latents [prompts, L] and the pre-drawn noise [steps, prompts, L], device fp32
    void* samp_noise = nullptr;
    size_t samp_x_bytes = 0, samp_noise_bytes = 0;

    osg_graph* graph = nullptr;
    Lowering* lowering = nullptr;  // kept alive: the launch closures capture it
    int runs = 0;
    bool in_capture = false;      // run_steps() is recording the hipGraph (no synchronisation allowed inside)
    double m_last_ms = 0;
    // options the plan was built with
    bool fp16 = true;
    bool u8_qdq = false, autotune = false, calibrate = false, fuse_attn = false, sdp_attn = false;
    std::set<std::string> outputs_convert_set;
    bool u8 = false;               // m_use_uint8_arithmetic: uint8 activations (the reference's W8A8 path, VAE decoder)
    bool stream_weights = false;
    bool fuse_ln_gemm = false;
    bool concat_views = true;     // m_hip_concat_views
    bool fuse_tblock = true;      // m_hip_fuse_tblock
    bool in_flight = false;       // a pass of this plan may still be running on the device (set while execute() / replay() are between enqueue and wait)
    bool gn_stats_on = false;
    int gn_stats_req = 2;          // m_hip_gn_stats as requested (0 off, 1 all eligible, 2 large tensors only)
    long gn_stats_min_elems = 0;   // m_hip_gn_stats: GroupNorm statistics from the producing convolutions' epilogues (plan.cpp lower_group_norm)
    char* gn_stats = nullptr;      // the pass's statistics block: one [N][G][2] int64 table per such GroupNorm, zeroed at the start of every pass
    size_t gn_stats_bytes = 0;
    void zero_gn_stats();
    void run_steps(size_t begin = 0, size_t end = (size_t)-1);   // steps [begin, end)
    // uint8 plans: steps [0, dyn_end) read values quantised per run (a pushed input and what merely re-arranges its codes): they run eagerly every
    // pass with the parameters of that pass, the steps after them only see range-data parameters and are captured like any other plan
    size_t dyn_end = 0;
    bool w8_resident = false;      // uint8 Conv/MatMul/Gemm weights kept as codes, dequantised inside the kernels (osg_*_w8)   // m_hip_stream_weights: weights are re-pulled from the WeightsProvider and re-streamed H2D every pass
    int fusion = 2;
    int fusion_req = 2;            // m_hip_fusion_level as the Model had it when the plan was built (build() lowers `fusion` to 0 for uint8 / calibration plans:
                                   // compatible() compares the REQUEST -- comparing the effective level re-planned every uint8 call, round 3)
    std::vector<std::string> extra_outputs;

    // ---- helpers used by the lowering code (plan.cpp) ----
    int new_val(const std::string& name, const std::vector<long>& shape, osg_dtype dt, Lay lay, bool batched);
    int alias(int v, const std::vector<long>& shape, Lay lay, const std::string& name = "");
    int root_of(int v) const;
    void* ptr(int v) const;
    size_t val_bytes(int v) const;
    long total_elems(int v) const;
    void add_step(const std::string& what, std::vector<int> reads, std::vector<int> writes, std::function<void()> fn);
    const Val& qv(int v) const;    // the val holding v's quantisation parameters
    void share_q(int dst, int src); // dst carries src's quantisation parameters
    int ensure_plain(int v);
    int ensure_nhwc(int v);
    // a device constant that survives this plan (ConstPool::derived) -- *fresh tells the caller to fill it; plan-owned (always fresh) in
    // streamed-weights mode or with an empty tag
    void* const_alloc(const std::string& tag, size_t bytes, bool* fresh);
    int ensure_dense(int v);   // materialise a strided column view as a dense tensor (for consumers that cannot take a leading dimension)
};

}  // namespace onnxstream
