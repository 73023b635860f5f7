// model.cpp -- Model front end: model.txt parser, tensor intake, range-data files, and run() which hands the graph to
// the device planner (plan.cpp).  The text format is the reference's (reference src/onnxstream.cpp:2445-2616):
//     name:Type*input:T;T;...*output:T;...[*attr:val;attr:val]
// with T = name(shape) for activations or file.bin(dtype:shape) for weights, dtype in
// float32|float16|int64|uint8[scale,zero_point].
#include "onnxstream.h"

#include <algorithm>
#include <cstring>

#include <thread>

#include "backend.h"
#include "plan.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace onnxstream {

std::string& trim(std::string& s) {
    const char* ws = " \t\n\r\f\v";
    s.erase(s.find_last_not_of(ws) + 1);
    s.erase(0, s.find_first_not_of(ws));
    return s;
}

namespace {

// split on a single-character-or-longer delimiter, keeping empty fields (an empty input slot is meaningful)
std::vector<std::string> split(const std::string& s, char delim) {
    std::vector<std::string> out;
    size_t b = 0;
    for (;;) {
        size_t e = s.find(delim, b);
        if (e == std::string::npos) {
            out.emplace_back(s.substr(b));
            return out;
        }
        out.emplace_back(s.substr(b, e - b));
        b = e + 1;
    }
}

}  // namespace

Model::Model(int threads_count) {
    m_backend_wanted = threads_count >= 0;
    // XnnPack::XnnPack(threads): 0 = one worker per hardware thread (reference :678-687)
    m_threads = threads_count > 0 ? (size_t)threads_count : std::max(1u, std::thread::hardware_concurrency());
}

Model::~Model() {
    m_data.clear();        // (device-resident tensors release their buffers while the backend is still there)
    delete m_plan;
    m_alive.reset();       // ... and copies of such tensors that outlive the Model must not touch it any more
    if (m_pool && m_backend) m_pool->clear(*m_backend);
    delete m_pool;
    delete m_backend;
}

void Model::hip_fetch_tensor(const std::string& name) {
    for (auto& t : m_data)
        if (t.m_name == name) {
            if (!t.m_hip_resident) return;   // already on the host
            if (!m_backend) throw std::runtime_error("Model::hip_fetch_tensor: no backend.");
            tensor_vector<uint16_t> host(t.m_hip_resident_bytes / 2);
            m_backend->check(m_backend->api.osg_download(m_backend->ctx, host.data(), t.m_hip_resident.get(), t.m_hip_resident_bytes), "osg_download");
            t.set_vector(std::move(host));
            t.m_hip_resident.reset();
            t.m_hip_resident_bytes = 0;
            return;
        }
    throw std::invalid_argument("Model::hip_fetch_tensor: tensor not found: " + name);
}

void Model::set_cuda_options(const CudaOptions& options) { m_cuda_options = options; }   // m_vram_to_use -> Plan's resident-weight budget; m_compute_fp32: accumulation is always fp32 here

void Model::read_file(const char* filename) {
    m_model = onnxstream::read_file<std::vector<char>>(filename);
    std::string fn(filename);
    auto sep = fn.find_last_of("/\\");
    m_path = sep == std::string::npos ? std::string() : fn.substr(0, sep + 1);
    get_wp()->m_path = m_path;
    m_ops_parsed = false;
    m_init_done = false;
    hip_invalidate_plan();
    if (m_pool && m_backend) m_pool->clear(*m_backend);   // another model: its resident weights go with the old one
}

void Model::read_string(const char* string, const char* path_with_slash) {
    m_model.assign(string, string + std::strlen(string));
    m_path = path_with_slash;
    get_wp()->m_path = m_path;
    m_ops_parsed = false;
    m_init_done = false;
    hip_invalidate_plan();
    if (m_pool && m_backend) m_pool->clear(*m_backend);
}

std::string Model::next_line() {
    const size_t n = m_model.size();
    while (m_pos < n && (m_model[m_pos] == '\r' || m_model[m_pos] == '\n')) m_pos++;
    const size_t start = m_pos;
    while (m_pos < n && m_model[m_pos] != '\r' && m_model[m_pos] != '\n') m_pos++;
    return std::string(m_model.data() + start, m_pos - start);
}

Tensor Model::parse_tensor_string(std::string& str) {
    Tensor t;
    if (str.empty()) return t;  // empty input slot (e.g. Resize roi)
    const auto open = str.find('(');
    if (open == std::string::npos || open == 0 || str.back() != ')' || str.find('(', open + 1) != std::string::npos ||
        str.size() < open + 2)
        throw std::invalid_argument("Model::parse_tensor_string: invalid tensor format.");
    t.m_name = str.substr(0, open);
    std::string inner = str.substr(open + 1, str.size() - open - 2);

    std::string shape = inner;
    const auto fields = split(inner, ':');
    if (fields.size() == 2) {
        const std::string& ty = fields[0];
        shape = fields[1];
        if (ty.rfind("uint8[", 0) == 0 && ty.back() == ']') {
            auto rng = split(ty.substr(6, ty.size() - 7), ',');
            if (rng.size() != 2) throw std::invalid_argument("Model::parse_tensor_string: invalid uint8 range.");
            t.m_type = TensorDataType::uint8;
            t.m_scale = (float)std::stod(rng[0]);
            t.m_zero_point = (uint8_t)std::stoi(rng[1]);
        } else if (ty == "float16") t.m_type = TensorDataType::float16;
        else if (ty == "float32") t.m_type = TensorDataType::float32;
        else if (ty == "int64") t.m_type = TensorDataType::int64;
        else throw std::invalid_argument("Model::parse_tensor_string: unsupported tensor data format.");
    } else if (fields.size() != 1) {
        throw std::invalid_argument("Model::parse_tensor_string: invalid tensor format.");
    }
    if (!shape.empty())
        for (auto& d : split(shape, ',')) {
            int i = std::stoi(d);
            if (i < 0) throw std::invalid_argument("Model::parse_tensor_string: invalid shape (dim < 0).");
            if (i == 0 && !m_support_dynamic_shapes) throw std::invalid_argument("Model::parse_tensor_string: invalid shape (dim == 0).");
            t.m_shape.push_back((size_t)i);
        }
    return t;
}

std::optional<Operation> Model::next_op_impl() {
    std::string line = next_line();
    if (line.empty()) return std::nullopt;
    auto sect = split(line, '*');
    if (sect.size() != 3 && sect.size() != 4) throw std::invalid_argument("Model::next_op: invalid format of model line.");
    Operation op;
    auto head = split(sect[0], ':');
    if (head.size() != 2) throw std::invalid_argument("Model::next_op: invalid format of model line.");
    op.m_name = head[0].empty() ? "onnxstream_fallback_name_" + std::to_string(m_pos) : head[0];
    op.m_type = head[1];
    if (sect[1].rfind("input:", 0) != 0 || sect[2].rfind("output:", 0) != 0)
        throw std::invalid_argument("Model::next_op: invalid format of model line.");
    for (auto& s : split(sect[1].substr(6), ';')) op.m_input.push_back(parse_tensor_string(s));
    for (auto& s : split(sect[2].substr(7), ';')) op.m_output.push_back(parse_tensor_string(s));
    if (sect.size() == 4)
        for (auto& kv : split(sect[3], ';')) {
            auto pair = split(kv, ':');
            if (pair.size() != 2) throw std::invalid_argument("Model::next_op: invalid format of model line.");
            op.m_attributes.emplace_back(pair[0], pair[1]);
        }
    return op;
}

void Model::parse_all() {
    if (m_ops_parsed) return;
    m_ops.clear();
    m_pos = 0;
    while (auto op = next_op_impl()) m_ops.push_back(std::move(*op));
    m_ops_parsed = true;
}

// First call: walk the graph once, count the consumers of every intermediate and announce every weight occurrence to
// the weights provider in model order (reference :3499-3548).  Later calls: tell the provider a new pass starts.
void Model::init() {
    if (!m_init_done) {
        parse_all();
        m_intermediate_refs.clear();
        for (auto& op : m_ops)
            for (auto& in : op.m_input) {
                if (in.m_name.empty()) continue;
                if (in.m_type == TensorDataType::none) {
                    m_intermediate_refs[in.m_name]++;
                } else {
                    size_t bytes = detail::dtype_size(in.m_type);
                    if (!bytes) throw std::invalid_argument("Model::run: unable to calculate tensor size: invalid type.");
                    get_wp()->on_init(in.m_type, in.m_name, in.element_count() * bytes);
                }
            }
        for (auto& name : m_extra_outputs) m_intermediate_refs[name]++;
        m_init_done = true;
    } else {
        m_first_run = false;
        if (m_hip_stream_weights || m_cuda_options.m_vram_to_use > 0) get_wp()->on_restart();
    }
}

void Model::push_tensor(Tensor&& t) {
    // the application pushes fp32 (or int64) host tensors; several pushes under one name chain into m_batch and are
    // executed as ONE batched pass (the reference executes each op N times, :3040-3050 + :3847)
    for (auto it = m_data.rbegin(); it != m_data.rend(); ++it)
        if (it->m_name == t.m_name) {
            if (!it->m_batch) it->m_batch = std::make_shared<std::vector<Tensor>>();
            it->m_batch->push_back(std::move(t));
            return;
        }
    m_data.push_back(std::move(t));
}

void Model::read_range_data(const char* filename) {
    auto file = onnxstream::read_file<std::vector<char>>(filename);
    size_t pos = 0;
    auto line_at = [&]() {
        while (pos < file.size() && (file[pos] == '\r' || file[pos] == '\n')) pos++;
        size_t s = pos;
        while (pos < file.size() && file[pos] != '\r' && file[pos] != '\n') pos++;
        return std::string(file.data() + s, pos - s);
    };
    for (;;) {
        std::string line = line_at();
        if (line.empty()) break;
        auto parts = split(line, ',');
        if (parts.size() != 3) throw std::invalid_argument("read_range_data: file format error.");
        m_range_data[parts[0]] = std::make_pair(std::stof(parts[1]), std::stof(parts[2]));
    }
}

void Model::write_range_data(const char* filename) {
    std::string text;
    for (auto& e : m_range_data) text += e.first + "," + std::to_string(e.second.first) + "," + std::to_string(e.second.second) + "\r\n";
    std::vector<char> v(text.begin(), text.end());
    onnxstream::write_file(filename, v);
}

void Model::hip_invalidate_plan() {
    delete m_plan;
    m_plan = nullptr;
}

void Model::hip_replay(int n, float* ms_each) {
    if (!m_plan) throw std::runtime_error("Model::hip_replay: no plan (call run() first).");
    m_plan->replay(n, ms_each);
    m_last_ms = m_plan->last_ms();
}

void Model::hip_set_input(const std::string& name, long index, const float* data, size_t count) {
    if (!m_plan) throw std::runtime_error("Model::hip_set_input: no plan (call run() first).");
    m_plan->set_input(name, index, data, count);
}

double Model::hip_sampler_loop(const std::string& sample_name, const std::string& timestep_name, const std::string& out_name, int steps, int prompts,
                               float* x, const float* noise, const float* c_in, const float* c_out, const float* t, const float* sigma, const float* d_sigma,
                               const float* sigma_up, float guidance, const float* clip) {
    if (!m_plan) throw std::runtime_error("Model::hip_sampler_loop: no plan (call run() first).");
    const double ms = m_plan->sampler_loop(sample_name, timestep_name, out_name, steps, prompts, x, noise, c_in, c_out, t, sigma, d_sigma, sigma_up, guidance, clip);
    m_last_ms = m_plan->last_ms();
    return ms;
}

std::string Model::hip_plan_info() const {
    if (!m_plan) throw std::runtime_error("Model::hip_plan_info: no plan (call run() first).");
    return m_plan->info();
}

std::string Model::hip_profile(int reps) {
    if (!m_plan) throw std::runtime_error("Model::hip_profile: no plan (call run() first).");
    return m_plan->profile(reps);
}

size_t Model::hip_streamed_bytes() const { return m_plan ? m_plan->streamed_bytes : 0; }
size_t Model::hip_resident_weight_bytes() const { return m_plan ? (m_plan->budgeted ? m_plan->resident_bytes + m_plan->ring_bytes : m_plan->weight_bytes) : 0; }

size_t Model::hip_last_kernel_count() const { return m_last_kernels; }
double Model::hip_last_pass_ms() const { return m_last_ms; }

void Model::run() {
    static const bool exec_times = std::getenv("OSG_EXEC_TIMES") != nullptr;
    const auto t_run = std::chrono::steady_clock::now();
    init();
    if (!m_backend_wanted) throw std::runtime_error("Model::run: this Model was created without a backend (threads_count < 0).");
    if (!m_backend) m_backend = new HipBackend(m_hip_device);

    // batch size = 1 + number of extra pushes, identical for every pushed input (reference :3817-3842)
    size_t batch = 1;
    for (auto& t : m_data) {
        size_t s = t.m_batch ? t.m_batch->size() + 1 : 1;
        if (s > 1) {
            if (batch > 1 && batch != s) throw std::invalid_argument("Model::run: inconsistent m_batch.size() across two or more tensors.");
            batch = s;
        }
    }
    static const bool timing = std::getenv("OSG_PLAN_TIMING") != nullptr;   // host milliseconds of a call's parts (the LLM flow re-plans on every call)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    // a plan that no longer fits is replaced -- and torn down only while the device works on the new plan's pass (Plan::execute's hook): its buffers are idle
    // (the pass it ran has been waited for), its pooled arena goes back to the pool for the call after this one
    // ... for models that re-plan on EVERY call (m_support_dynamic_shapes: the LLM flow -- small arenas that come from and go back to the Model's pool, 0.8 ms of
    // teardown per token).  Any other model (the SD UNet / VAE: a re-plan is a batch, option or budget change) gives its arena, streaming ring and owned
    // buffers back BEFORE the replacement is built: old and new arena never coexist, m_vram_to_use is not exceeded (advisor, round 3)
    Plan* old_plan = nullptr;
    if (m_plan && !m_plan->compatible(*this, batch)) {
        if (m_plan->recycle) old_plan = m_plan;
        else delete m_plan;
        m_plan = nullptr;
    }
    struct OldPlan {
        Plan*& p;
        ~OldPlan() { delete p; p = nullptr; }       // (whatever happens below)
    } old_guard{old_plan};
    const auto t1 = now();
    const bool rebuilt = !m_plan;
    if (!m_plan) {
        if (!m_pool) m_pool = new ConstPool();
        m_plan = new Plan(*this, *m_backend, *m_pool, batch);
        m_plans_built++;
        try {
            m_plan->build();
        } catch (...) {
            hip_invalidate_plan();
            throw;
        }
    }
    const auto t2 = now();
    m_plan->execute([&] {
        delete old_plan;
        old_plan = nullptr;
    });
    if (timing && rebuilt)
        fprintf(stderr, "[run] drop the old plan %.2f ms, new plan %.2f ms, execute %.2f ms (gathered transfers: %zu B up, %zu B down)\n", ms(t0, t1), ms(t1, t2), ms(t2, now()),
                m_plan->gathered_up, m_plan->gathered_down);
    m_last_kernels = m_plan->kernel_count();
    m_last_ms = m_plan->last_ms();
    if (exec_times) fprintf(stderr, "[run] before execute %.3f ms, whole call %.3f ms\n", ms(t_run, t2), ms(t_run, now()));
}

}  // namespace onnxstream
