// onnxstream.h -- host-side API of the MI355X backend, source-compatible with the reference's public surface
// (/root/reference/src/onnxstream.h): namespace onnxstream, tensor_vector / Tensor / Operation (:22-264), the
// WeightsProvider family (:266-900) and class Model with its option fields (:913-968).  The reference's
// src/exports.cpp compiles UNCHANGED against this header (tests/test_dropin_cpu.py does exactly that), which is the
// drop-in contract.  Everything behind the API is new: Model::run() does not interpret ops on the host -- it lowers
// the whole graph once into a plan of HIP kernel launches over device-resident activations (see plan.h) and replays it.
#pragma once

#include <any>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <variant>
#include <vector>

namespace onnxstream {

// XNNPACK kernels may over-read a few bytes; the reference pads every allocation by 16 bytes (XNN_EXTRA_BYTES) and
// callers of the C API rely on that slack too.  Kept for ABI-level compatibility of buffers handed to user code.
#define TENSOR_VECTOR_EXTRA_BYTES 16

template <class T>
struct Mallocator {
    using value_type = T;
    Mallocator() = default;
    template <class U>
    constexpr Mallocator(const Mallocator<U>&) noexcept {}
    [[nodiscard]] T* allocate(std::size_t n) {
        if (n > (std::numeric_limits<std::size_t>::max() - TENSOR_VECTOR_EXTRA_BYTES) / sizeof(T)) throw std::bad_array_new_length();
        void* p = std::malloc(n * sizeof(T) + TENSOR_VECTOR_EXTRA_BYTES);
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, std::size_t) noexcept { std::free(p); }
};
template <class T, class U> bool operator==(const Mallocator<T>&, const Mallocator<U>&) { return true; }
template <class T, class U> bool operator!=(const Mallocator<T>&, const Mallocator<U>&) { return false; }

template <class T> using tensor_vector = std::vector<T, Mallocator<T>>;
template <typename T> tensor_vector<T> create_tensor_vector(size_t size) { return tensor_vector<T>(size); }

class scope_guard {
public:
    template <class F> explicit scope_guard(F&& f) : m_f(std::forward<F>(f)) {}
    ~scope_guard() { if (m_active && m_f) m_f(); }
    scope_guard(const scope_guard&) = delete;
    scope_guard& operator=(const scope_guard&) = delete;
    bool m_active = true;
private:
    std::function<void()> m_f;
};

template <typename V>
V read_file(const char* filename) {
    using E = typename V::value_type;
    std::ifstream f(filename, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("read_file: unable to open file (" + std::string(filename) + ").");
    const std::streamoff bytes = f.tellg();
    if (bytes <= 0 || bytes % (std::streamoff)sizeof(E)) throw std::invalid_argument("read_file: invalid size of file.");
    V out((size_t)bytes / sizeof(E));
    f.seekg(0);
    if (!f.read(reinterpret_cast<char*>(out.data()), bytes)) throw std::runtime_error("read_file: unable to read file.");
    return out;
}

template <typename V>
void write_file(const char* filename, const V& data) {
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("write_file: unable to open file.");
    if (!f.write(reinterpret_cast<const char*>(data.data()), data.size() * sizeof(typename V::value_type)))
        throw std::runtime_error("write_file: unable to write file.");
}

std::string& trim(std::string& s);

enum class TensorDataType { none, uint8, float16, float32, int64 };
enum class TensorDataLayout { unspecified, nhwc };

namespace detail {
template <typename T> struct dtype_of;
template <> struct dtype_of<uint8_t> { static constexpr TensorDataType value = TensorDataType::uint8; };
template <> struct dtype_of<uint16_t> { static constexpr TensorDataType value = TensorDataType::float16; };
template <> struct dtype_of<float> { static constexpr TensorDataType value = TensorDataType::float32; };
template <> struct dtype_of<int64_t> { static constexpr TensorDataType value = TensorDataType::int64; };

inline size_t dtype_size(TensorDataType t) {
    switch (t) {
        case TensorDataType::uint8: return 1;
        case TensorDataType::float16: return 2;
        case TensorDataType::float32: return 4;
        case TensorDataType::int64: return 8;
        default: return 0;
    }
}

// calls f(type_tag<T>) for the element type matching `t`
template <typename T> struct type_tag { using type = T; };
template <typename F>
decltype(auto) dispatch_dtype(TensorDataType t, F&& f, const char* what = "unsupported tensor data format.") {
    switch (t) {
        case TensorDataType::uint8: return f(type_tag<uint8_t>{});
        case TensorDataType::float16: return f(type_tag<uint16_t>{});
        case TensorDataType::float32: return f(type_tag<float>{});
        case TensorDataType::int64: return f(type_tag<int64_t>{});
        default: throw std::invalid_argument(what);
    }
}
}  // namespace detail

class Tensor {
public:
    std::string m_name;
    TensorDataType m_type = TensorDataType::none;
    std::vector<size_t> m_shape;
    std::variant<std::shared_ptr<tensor_vector<uint8_t>>, std::shared_ptr<tensor_vector<uint16_t>>,
                 std::shared_ptr<tensor_vector<float>>, std::shared_ptr<tensor_vector<int64_t>>>
        m_data;
    TensorDataLayout m_layout = TensorDataLayout::unspecified;
    float m_scale = 0;
    uint8_t m_zero_point = 0;
    bool m_is_static_weights = false;
    std::shared_ptr<std::vector<Tensor>> m_batch;  // further samples pushed under the same name (reference :3040-3050)
    // backend addition (Model::m_hip_resident_outputs): this fp16 tensor's data is a DEVICE buffer (the host vector is empty) -- an output that never left
    // the GPU.  Pushed back under an input name (llm.cpp's opkv* -> pkv* renaming) it is read where it lies; Model::hip_fetch_tensor brings it to the host.
    std::shared_ptr<void> m_hip_resident;
    size_t m_hip_resident_bytes = 0;

    template <typename T>
    tensor_vector<T>& get_vector() {
        if (m_type != detail::dtype_of<T>::value) throw std::invalid_argument("Tensor::get_vector: invalid type.");
        return *std::get<std::shared_ptr<tensor_vector<T>>>(m_data);
    }
    template <typename T>
    void set_vector(tensor_vector<T>&& v) {
        m_type = detail::dtype_of<T>::value;
        m_data = std::make_shared<tensor_vector<T>>(std::move(v));
    }
    void make_copy_of_data() {
        detail::dispatch_dtype(m_type, [&](auto tag) {
            using T = typename decltype(tag)::type;
            set_vector(tensor_vector<T>(get_vector<T>()));
        }, "Tensor::make_copy_of_data: invalid type.");
    }
    size_t element_count() const {
        size_t n = 1;
        for (auto d : m_shape) n *= d;
        return n;
    }
};

class Operation {
public:
    std::string m_name;
    std::string m_type;
    std::vector<Tensor> m_input;
    std::vector<Tensor> m_output;
    std::vector<std::pair<std::string, std::string>> m_attributes;
};

// ---------------------------------------------------------------------------------------------------------------------
// Weights providers.  Contract (reference :266-291 and the call sites in Model): on_init(type,name,bytes) once per
// weight occurrence in strict model order during the first init(); get_<type>(name) / getptr_<type>(name) in that same
// order; remove(name) when the runtime has made the weight resident; on_restart() before every later pass.
// ---------------------------------------------------------------------------------------------------------------------
class WeightsProvider {
public:
    WeightsProvider() {}
    virtual ~WeightsProvider() {}
    std::string m_path;

    virtual void on_init(TensorDataType, const std::string&, size_t) {}
    virtual void on_restart() {}
    virtual void remove(const std::string&) {}
    virtual void update(Tensor&) {}
    virtual TensorDataType get_type_of_next() { return TensorDataType::none; }

    virtual bool supports_getptr() { return false; }
    virtual std::shared_ptr<tensor_vector<uint8_t>> getptr_uint8(const std::string&) { throw std::invalid_argument("getptr not supported."); }
    virtual std::shared_ptr<tensor_vector<uint16_t>> getptr_float16(const std::string&) { throw std::invalid_argument("getptr not supported."); }
    virtual std::shared_ptr<tensor_vector<float>> getptr_float32(const std::string&) { throw std::invalid_argument("getptr not supported."); }
    virtual std::shared_ptr<tensor_vector<int64_t>> getptr_int64(const std::string&) { throw std::invalid_argument("getptr not supported."); }

    virtual tensor_vector<uint8_t> get_uint8(const std::string& name) = 0;
    virtual tensor_vector<uint16_t> get_float16(const std::string& name) = 0;
    virtual tensor_vector<float> get_float32(const std::string& name) = 0;
    virtual tensor_vector<int64_t> get_int64(const std::string& name) = 0;
};

// Dry-run listing of the weights a model needs (reference :293-329).
class CollectNamesWeightsProvider : public WeightsProvider {
public:
    struct Entry {
        TensorDataType m_type;
        std::string m_name;
        size_t m_size;
        Entry(TensorDataType type, const std::string& name, size_t size) : m_type(type), m_name(name), m_size(size) {}
    };
    bool m_use_vector = false;
    std::set<std::string> m_names;
    std::vector<Entry> m_names_vec;

    CollectNamesWeightsProvider(bool use_vector = false) : m_use_vector(use_vector) {}
    void on_init(TensorDataType type, const std::string& name, size_t size) override {
        if (m_use_vector) m_names_vec.emplace_back(type, name, size);
        else m_names.insert(name);
    }
    tensor_vector<uint8_t> get_uint8(const std::string&) override { throw std::invalid_argument("Not implemented."); }
    tensor_vector<uint16_t> get_float16(const std::string&) override { throw std::invalid_argument("Not implemented."); }
    tensor_vector<float> get_float32(const std::string&) override { throw std::invalid_argument("Not implemented."); }
    tensor_vector<int64_t> get_int64(const std::string&) override { throw std::invalid_argument("Not implemented."); }
};

// Reads each file at the moment it is requested (reference :331-354).
class DiskNoCacheWeightsProvider : public WeightsProvider {
public:
    tensor_vector<uint8_t> get_uint8(const std::string& n) override { return read_file<tensor_vector<uint8_t>>((m_path + n).c_str()); }
    tensor_vector<uint16_t> get_float16(const std::string& n) override { return read_file<tensor_vector<uint16_t>>((m_path + n).c_str()); }
    tensor_vector<float> get_float32(const std::string& n) override { return read_file<tensor_vector<float>>((m_path + n).c_str()); }
    tensor_vector<int64_t> get_int64(const std::string& n) override { return read_file<tensor_vector<int64_t>>((m_path + n).c_str()); }
};

// Background reader with a bounded look-ahead (reference :356-664: one worker thread, look-ahead limited to
// `max_memory` bytes "plus one file", strict on_init order, errors of the worker re-thrown on the consumer side).
// This implementation keeps a FIFO of pending entries and a sliding window of loaded ones; the consumer blocks on a
// condition variable instead of polling.
class DiskPrefetchWeightsProvider : public WeightsProvider {
    using Blob = std::variant<std::monostate, tensor_vector<uint8_t>, tensor_vector<uint16_t>, tensor_vector<float>, tensor_vector<int64_t>>;
    struct Item {
        TensorDataType type;
        std::string name;
        size_t bytes;
    };
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<std::pair<Item, Blob>> ready;  // loaded, in order
        size_t ready_bytes = 0;
        size_t next = 0;                          // index into `order` of the next file to read
        size_t done = 0;                          // files read AND published in `ready` so far (the file `next` points past may still be loading)
        size_t total = 0;                         // length of the order snapshot the worker runs on (m_order shrinks under remove() DURING a pass)
        bool stop = false;
        std::string error;
    };

    size_t m_max_memory;
    bool m_limit_plus_one_file;
    std::vector<Item> m_order;       // model order, fixed after the first pass (minus remove()d names)
    bool m_order_frozen = false;
    std::shared_ptr<Shared> m_sh;
    std::thread m_thread;

    void stop_worker() {
        if (m_thread.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m_sh->mu);
                m_sh->stop = true;
            }
            m_sh->cv.notify_all();
            m_thread.join();
        }
        m_sh.reset();
    }

    static Blob load(const std::string& path, const Item& it) {
        return detail::dispatch_dtype(it.type, [&](auto tag) -> Blob {
            using T = typename decltype(tag)::type;
            return read_file<tensor_vector<T>>((path + it.name).c_str());
        }, "DiskPrefetchWeightsProvider::worker: invalid type.");
    }

    void worker(std::shared_ptr<Shared> sh, std::string path, std::vector<Item> order) {
        try {
            if (!m_max_memory) throw std::invalid_argument("DiskPrefetchWeightsProvider::worker: m_max_memory cannot be zero.");
            for (;;) {
                Item it;
                {
                    std::unique_lock<std::mutex> lk(sh->mu);
                    sh->cv.wait(lk, [&] {
                        if (sh->stop || sh->next >= order.size()) return true;
                        // window rule: keep reading while loaded bytes < budget; with "plus one file" one more file may
                        // be read after the budget has been reached
                        if (sh->ready_bytes < m_max_memory) return true;
                        if (m_limit_plus_one_file && sh->ready.size() <= 1) return true;
                        return false;
                    });
                    if (sh->stop || sh->next >= order.size()) return;
                    it = order[sh->next++];
                }
                Blob b = load(path, it);
                {
                    std::lock_guard<std::mutex> lk(sh->mu);
                    sh->ready_bytes += it.bytes;
                    sh->ready.emplace_back(std::move(it), std::move(b));
                    sh->done++;
                }
                sh->cv.notify_all();
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(sh->mu);
            sh->error = e.what();
            sh->cv.notify_all();
        }
    }

    template <typename T>
    tensor_vector<T> provide(const std::string& name) {
        if (!m_sh) {
            m_order_frozen = true;
            m_sh = std::make_shared<Shared>();
            m_sh->total = m_order.size();
            m_thread = std::thread(&DiskPrefetchWeightsProvider::worker, this, m_sh, m_path, m_order);
        }
        std::unique_lock<std::mutex> lk(m_sh->mu);
        // (exhausted = everything PUBLISHED and consumed; `next` alone runs ahead of the file the worker is still reading)
        m_sh->cv.wait(lk, [&] { return !m_sh->ready.empty() || !m_sh->error.empty() || m_sh->done >= m_sh->total; });
        if (!m_sh->error.empty())
            throw std::invalid_argument("DiskPrefetchWeightsProvider::provide: fatal error in worker thread: \"" + m_sh->error + "\".");
        if (m_sh->ready.empty()) throw std::invalid_argument("DiskPrefetchWeightsProvider::provide: vector is empty.");
        auto& front = m_sh->ready.front();
        if (front.first.name != name) throw std::invalid_argument("DiskPrefetchWeightsProvider::provide: invalid name.");
        if (!std::holds_alternative<tensor_vector<T>>(front.second))
            throw std::invalid_argument("DiskPrefetchWeightsProvider::provide: invalid type.");
        tensor_vector<T> v = std::move(std::get<tensor_vector<T>>(front.second));
        if (v.empty()) throw std::invalid_argument("DiskPrefetchWeightsProvider::provide: invalid size.");
        m_sh->ready_bytes -= front.first.bytes;
        m_sh->ready.pop_front();
        lk.unlock();
        m_sh->cv.notify_all();
        return v;
    }

    static std::string disk_name(const std::string& name) {
        auto pos = name.find("_nchw.bin");
        return pos == std::string::npos ? name : name.substr(0, pos) + "_nhwc.bin";
    }

public:
    DiskPrefetchWeightsProvider(size_t max_memory = 1 * 1024 * 1024, bool limit_plus_one_file = true)
        : m_max_memory(max_memory), m_limit_plus_one_file(limit_plus_one_file) {}
    DiskPrefetchWeightsProvider(DiskPrefetchWeightsProvider&& o) noexcept
        : WeightsProvider(o), m_max_memory(o.m_max_memory), m_limit_plus_one_file(o.m_limit_plus_one_file),
          m_order(std::move(o.m_order)), m_order_frozen(o.m_order_frozen) {
        o.stop_worker();
    }
    DiskPrefetchWeightsProvider(const DiskPrefetchWeightsProvider& o)
        : WeightsProvider(o), m_max_memory(o.m_max_memory), m_limit_plus_one_file(o.m_limit_plus_one_file), m_order(o.m_order),
          m_order_frozen(o.m_order_frozen) {}
    ~DiskPrefetchWeightsProvider() override { stop_worker(); }

    void on_init(TensorDataType type, const std::string& name, size_t size) override {
        if (!m_order_frozen) m_order.push_back(Item{type, disk_name(name), size});
    }
    void on_restart() override { stop_worker(); }
    void remove(const std::string& name) override {
        for (size_t i = 0; i < m_order.size(); i++)
            if (m_order[i].name == name) {
                m_order.erase(m_order.begin() + i);
                return;
            }
        throw std::invalid_argument("DiskPrefetchWeightsProvider::remove: name not found.");
    }
    tensor_vector<uint8_t> get_uint8(const std::string& n) override { return provide<uint8_t>(n); }
    tensor_vector<uint16_t> get_float16(const std::string& n) override { return provide<uint16_t>(n); }
    tensor_vector<float> get_float32(const std::string& n) override { return provide<float>(n); }
    tensor_vector<int64_t> get_int64(const std::string& n) override { return provide<int64_t>(n); }
};

// Keeps every weight in host RAM (reference :666-900): on the first pass it pulls from the wrapped reader `R` and
// records the sequence; later passes replay the recorded sequence by index and hand out shared pointers (zero copy).
// Constructed without a reader it is filled by the application through add_empty_and_return_ptr().
template <typename R>
class RamWeightsProvider : public WeightsProvider {
    using Ptr = std::variant<std::monostate, std::shared_ptr<tensor_vector<uint8_t>>, std::shared_ptr<tensor_vector<uint16_t>>,
                             std::shared_ptr<tensor_vector<float>>, std::shared_ptr<tensor_vector<int64_t>>>;
    struct Slot {
        std::string m_name;
        Ptr m_data;
    };
    std::shared_ptr<R> m_reader;
    bool m_reader_specified = true;
    bool m_path_forwarded = false;
    std::vector<Slot> m_weights;
    size_t m_cursor = 0;

    template <typename U>
    std::shared_ptr<tensor_vector<U>> provide(const std::string& name) {
        using P = std::shared_ptr<tensor_vector<U>>;
        if (m_reader) {
            if (!m_path_forwarded) {
                m_reader->m_path = m_path;
                m_path_forwarded = true;
            }
            tensor_vector<U> v;
            if constexpr (std::is_same_v<U, uint8_t>) v = m_reader->get_uint8(name);
            else if constexpr (std::is_same_v<U, uint16_t>) v = m_reader->get_float16(name);
            else if constexpr (std::is_same_v<U, float>) v = m_reader->get_float32(name);
            else v = m_reader->get_int64(name);
            P p = std::make_shared<tensor_vector<U>>(std::move(v));
            m_weights.push_back(Slot{name, p});
            return p;
        }
        if (m_cursor >= m_weights.size()) throw std::invalid_argument("RamWeightsProvider::provide: invalid index.");
        Slot& s = m_weights[m_cursor++];
        if (s.m_name != name)
            throw std::invalid_argument("RamWeightsProvider::provide: invalid name (requested: '" + name + "', available: '" + s.m_name + "').");
        if (!std::holds_alternative<P>(s.m_data)) throw std::invalid_argument("RamWeightsProvider::provide: invalid data type.");
        return std::get<P>(s.m_data);
    }

public:
    RamWeightsProvider() { m_reader_specified = false; }
    RamWeightsProvider(R&& reader) { m_reader = std::make_shared<R>(std::move(reader)); }
    ~RamWeightsProvider() override {}

    void on_init(TensorDataType type, const std::string& name, size_t size) override {
        if (m_reader) m_reader->on_init(type, name, size);
        else if (m_reader_specified) throw std::invalid_argument("RamWeightsProvider::on_init: invalid call to on_init.");
    }
    void on_restart() override {
        m_reader.reset();
        m_cursor = 0;
    }
    void remove(const std::string& name) override {
        for (size_t i = 0; i < m_weights.size(); i++)
            if (m_weights[i].m_name == name) {
                m_weights.erase(m_weights.begin() + i);
                if (!m_reader_specified && m_cursor > i) m_cursor--;
                return;
            }
        throw std::invalid_argument("RamWeightsProvider::remove: name not found.");
    }
    void update(Tensor& t) override {
        for (auto& s : m_weights)
            if (s.m_name == t.m_name) {
                detail::dispatch_dtype(t.m_type, [&](auto tag) {
                    using T = typename decltype(tag)::type;
                    s.m_data = std::make_shared<tensor_vector<T>>(t.get_vector<T>());
                });
                return;
            }
        throw std::invalid_argument("RamWeightsProvider::update: name not found.");
    }
    TensorDataType get_type_of_next() override {
        if (m_reader) return TensorDataType::none;
        if (m_cursor >= m_weights.size()) throw std::invalid_argument("RamWeightsProvider::get_type_of_next: invalid index.");
        switch (m_weights[m_cursor].m_data.index()) {
            case 1: return TensorDataType::uint8;
            case 2: return TensorDataType::float16;
            case 3: return TensorDataType::float32;
            case 4: return TensorDataType::int64;
        }
        throw std::invalid_argument("RamWeightsProvider::get_type_of_next: unable to determine type.");
    }

    tensor_vector<uint8_t> get_uint8(const std::string& n) override { return *provide<uint8_t>(n); }
    tensor_vector<uint16_t> get_float16(const std::string& n) override { return *provide<uint16_t>(n); }
    tensor_vector<float> get_float32(const std::string& n) override { return *provide<float>(n); }
    tensor_vector<int64_t> get_int64(const std::string& n) override { return *provide<int64_t>(n); }

    bool supports_getptr() override { return true; }
    std::shared_ptr<tensor_vector<uint8_t>> getptr_uint8(const std::string& n) override { return provide<uint8_t>(n); }
    std::shared_ptr<tensor_vector<uint16_t>> getptr_float16(const std::string& n) override { return provide<uint16_t>(n); }
    std::shared_ptr<tensor_vector<float>> getptr_float32(const std::string& n) override { return provide<float>(n); }
    std::shared_ptr<tensor_vector<int64_t>> getptr_int64(const std::string& n) override { return provide<int64_t>(n); }

    template <typename U>
    void* add_empty_and_return_ptr(const std::string& name, size_t size) {
        auto p = std::make_shared<tensor_vector<U>>(size);
        m_weights.push_back(Slot{name, p});
        return p->data();
    }
};

class XnnPack;  // kept only so that code naming the type still compiles; the device backend replaces it
class HipBackend;
struct Plan;
struct Lowering;
struct ConstPool;

// The reference's CudaOptions (:904-911) -- accepted and mapped onto the HIP backend: vram budget => how many bytes of
// weights may stay resident before the runtime falls back to streaming them every pass.
struct CudaOptions {
    uint64_t m_vram_to_use = 0;
    bool m_compute_fp32 = false;
    CudaOptions() {}
    CudaOptions(uint64_t vram_to_use, bool compute_fp32) : m_vram_to_use(vram_to_use), m_compute_fp32(compute_fp32) {}
};

class Model {
public:
    Model(int threads_count = 0);  // threads_count < 0: no backend is created (dry runs, e.g. CollectNames)
    ~Model();
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;

    template <typename T>
    void set_weights_provider(T&& wp) {
        if (m_wp_interface_internal) throw std::invalid_argument("Model::set_weights_provider: weights provider already set.");
        m_wp_object = std::move(wp);
        m_wp_interface_internal = std::any_cast<T>(&m_wp_object);
    }
    template <typename T>
    T& get_weights_provider() { return *(T*)get_wp(); }

    void read_file(const char* filename);
    void read_string(const char* string, const char* path_with_slash = "./");

    std::vector<Tensor> m_data;  // inputs before run(); outputs (fp32, NCHW) after run()

    void push_tensor(Tensor&& t);
    void init();
    void run();

    void read_range_data(const char* filename);
    void write_range_data(const char* filename);
    std::map<std::string, std::pair<float, float>> m_range_data;
    bool m_range_data_calibrate = false;

    // ---- the reference's option fields (:949-968) ----
    bool m_use_fp16_arithmetic = false;
    bool m_use_uint8_qdq = false;
    bool m_use_uint8_arithmetic = false;
    bool m_fuse_ops_in_attention = false;
    size_t m_attention_fused_ops_parts = 2;  // accepted; the fused device attention never materialises score slabs
    std::vector<std::string> m_extra_outputs;
    bool m_force_fp16_storage = false;
    std::set<std::string> m_force_uint8_storage_set;
    bool m_support_dynamic_shapes = false;
    bool m_use_ops_cache = false;            // resident weights: host copies are remove()d from the provider
    std::function<bool(const std::string&, const std::string&)> m_requires_upcast;
    bool m_use_scaled_dp_attn_op = false;
    std::set<std::string> m_outputs_convert_set;
    bool m_use_next_op_cache = false;        // the parsed graph is always cached here
    bool m_use_nchw_convs = false;

    void set_cuda_options(const CudaOptions& options);

    bool m_ops_printf = false;
    bool m_ops_times_printf = false;

    bool is_model_empty() { return m_model.size() == 0; }

    // ---- additions of the MI355X backend (not in the reference) ----
    int m_hip_device = 0;          // HIP device ordinal this Model runs on (one Model per GPU / rank)
    int m_hip_fusion_level = 2;    // 0: one kernel per graph op (faithful roundings); 1: + elementwise/norm fusions; 2: + attention/epilogues
    bool m_hip_use_graph = true;   // replay a pass as one hipGraph from the third run on
    bool m_hip_fuse_ln_gemm = true;  // fusion level 2: a LayerNorm whose only consumers are Linear ops is folded into their GEMM (osg_gemm_ln, row
                                     // statistics handed over by the producing GEMM): 48 launches less in the SD 1.5 UNet, measured -1 % per step (round 2) => on
    int m_hip_gn_stats = 2;         // fusion level 2: a GroupNorm over what convolutions store reads its statistics from their epilogues (osg_set_stat_sinks) and is one streaming
                                    // launch.  0 off, 1 every eligible GroupNorm, 2 (default) where the tensor has >= 8 M elements: in the throughput regime it pays (f16 VAE decoder
                                    // 4.61 -> 4.08 ms, SDXL UNet -1.1 %), on the SD 1.5 pass -- launches that last as long as one workgroup -- it is neutral (profiles/r03_gn_stats_ab.txt)
    bool m_hip_fuse_tblock = true;  // fusion level 2: the row-local tail of a transformer block (attn1.to_out .. ff.net.2 [.. proj_out]) as ONE launch where osg_tblock_tail takes the shape (round 4)
    bool m_hip_concat_views = true; // fusion level 2: convolutions store skip tensors straight into their Concat slot (osg_conv2d_nhwc_v), no copy launch
    bool m_hip_autotune = false;   // true: the first (eager) pass TIMES the legal tile / split-K configurations of every GEMM / convolution shape
                                   // (osg_set_autotune) -- faster, but the choice (hence the fp32 summation order, hence the last bits) depends on a
                                   // timer unless OSG_TUNE_CACHE seeds it; default: the deterministic cost-model choice
    // true (with m_support_dynamic_shapes and a non-empty m_outputs_convert_set): the outputs that are NOT converted to fp32 stay on the device (see
    // Tensor::m_hip_resident) instead of being downloaded into m_data -- the key/value caches of the LLM flow then never cross PCIe
    bool m_hip_resident_outputs = false;
    void hip_fetch_tensor(const std::string& name);   // downloads a device-resident tensor of m_data into its host vector
    bool m_hip_w8_resident = false;     // true: uint8 weights of Conv/MatMul/Gemm (K % 64 == 0) stay uint8 CODES in HBM (half the footprint and weight traffic) and become
                                        // halves between the LDS tile and the MFMA of the tuned kernels (round 6, osg_gemm_w8.hip: exact q - zp into the MFMA, the scale on
                                        // the f32 accumulator -- the reference rounds each dequantised weight to f16 first, :2887-2891 / :3353); false (default): dequantise
                                        // once at load, the reference's order -- the f16 plan with its folded LayerNorms and fused block tails, 8-10 % faster
    bool m_hip_stream_weights = false;  // true: weights are re-streamed through pinned buffers every pass (WeightsProvider mode)
    size_t hip_last_kernel_count() const;
    size_t hip_plans_built() const { return m_plans_built; }   // how many times run() had to (re-)plan since the Model was created
    double hip_last_pass_ms() const;   // device time of the last pass (HIP events on the compute stream)
    void hip_invalidate_plan();
    size_t hip_streamed_bytes() const;
    size_t hip_resident_weight_bytes() const;   // device bytes held for weights: everything, or (VRAM budget) the resident part + the streaming ring
    void hip_replay(int n, float* ms_each);   // relaunch the captured pass on resident inputs (per-launch device ms)
    void hip_set_input(const std::string& name, long index, const float* data, size_t count);   // refresh one pushed sample of a resident input
    // the denoising loop with CFG + Euler-Ancestral on the device (see Plan::sampler_loop); returns the loop's device time in ms
    double hip_sampler_loop(const std::string& sample_name, const std::string& timestep_name, const std::string& out_name, int steps, int prompts,
                            float* x, const float* noise, const float* c_in, const float* c_out, const float* t, const float* sigma, const float* d_sigma,
                            const float* sigma_up, float guidance, const float* clip);
    std::string hip_plan_info() const;         // steps / arena placement of the current plan (see Plan::info)
    std::string hip_profile(int reps);         // per-step HIP-event timing report of an eager pass

private:
    friend struct Plan;
    friend struct Lowering;
    std::set<std::string> m_weights_exclusion_set;
    bool m_first_run = true;

    std::any m_wp_object;
    WeightsProvider* m_wp_interface_internal = nullptr;
    WeightsProvider* get_wp() {
        if (!m_wp_interface_internal) set_weights_provider(DiskPrefetchWeightsProvider());
        return m_wp_interface_internal;
    }

    std::string next_line();
    Tensor parse_tensor_string(std::string& str);
    std::optional<Operation> next_op_impl();
    void parse_all();

    std::vector<char> m_model;
    size_t m_pos = 0;
    std::string m_path;
    std::vector<Operation> m_ops;          // the whole parsed graph
    bool m_ops_parsed = false;
    std::map<std::string, int> m_intermediate_refs;
    bool m_init_done = false;

    CudaOptions m_cuda_options;
    HipBackend* m_backend = nullptr;       // takes the seat of `XnnPack* m_xnnpack` (reference :1036)
    bool m_backend_wanted = true;
    size_t m_threads = 1;                  // the reference's pthreadpool size: only the chunking of get_percentiles depends on it (qu8.h)
    Plan* m_plan = nullptr;
    ConstPool* m_pool = nullptr;           // device-resident weights, kept across plan rebuilds (plan.h)
    std::shared_ptr<bool> m_alive = std::make_shared<bool>(true);   // what the deleters of device-resident tensors check before they touch m_backend
    size_t m_last_kernels = 0;
    size_t m_plans_built = 0;
    double m_last_ms = 0;
};

}  // namespace onnxstream
