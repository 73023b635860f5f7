// plan_run.cpp -- the RUN-TIME half of Plan (round 6: split out of plan.cpp, which keeps construction, fusion and lowering): replaying the steps eagerly or as a
// captured hipGraph, input upload / output download, weight re-streaming under a VRAM budget, the device sampler loop, the per-step profile.
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
