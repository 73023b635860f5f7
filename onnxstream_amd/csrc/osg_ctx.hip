// libosgpu: context, memory, transfers, graph capture, timing.  (C ABI: include/osgpu.h)
#include "osg_common.h"
#include "osg_gemm_common.h"
#include "osg_tune.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <set>

namespace osg_tune {
static std::mutex g_mu;
static std::map<Key, Choice> g_table;
static bool g_loaded = false;
// OSG_TUNE_CACHE=<file>: measured choices persist across processes (text, one "key... choice..." line per shape, appended as they are
// made) -- a second process starts tuned, makes the same choices (bit-identical results run to run) and issues no timing launches.
static const char* cache_path() { return getenv("OSG_TUNE_CACHE"); }
static void load_locked() {
    if (g_loaded) return;
    g_loaded = true;
    const char* path = cache_path();
    if (!path) return;
    FILE* f = fopen(path, "r");
    if (!f) return;
    Key k{};
    Choice c{};
    while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %f", &k.kind, &k.device, &k.M, &k.N, &k.K, &k.batch, &k.H, &k.W, &k.Cin, &k.KW,
                  &k.sh, &k.sw, &k.flags, &c.family, &c.cfg, &c.nst, &c.splits, &c.bn, &c.us) == 19) {
        // a row that names no launchable configuration is dropped (the launchers would silently fall back to another tile, or divide by
        // splits == 0); the device ordinal is not part of the identity of a shape -- one table serves every rank of a node
        const int tile = c.cfg & 7, ks2 = c.cfg & 8, fold = c.cfg & 16, spec = c.cfg & 32;   /* bit 3 of cfg: KS = 2 (64x64 with 2 / 4 stages, 128x64 with 2); bit 4: split-K folded in the kernel; bit 5: four loader waves */
        bool ok;
        if (c.family == 0) {
            ok = c.cfg >= 0 && (c.cfg & ~63) == 0 && c.splits >= 1 && c.splits <= 64;
            ok = ok && (!spec || ((tile == 0 || tile == 4) && c.nst == 4 && !ks2 && !fold && c.splits == 1));
            ok = ok && (!fold || (!ks2 && tile != 0 && tile != 4 && tile != 7 && c.splits >= 2 && c.splits <= 4));
            ok = ok && (!ks2 || (tile == 2 && (c.nst == 2 || c.nst == 4)) || (tile == 1 && c.nst == 2));
            // = the instantiations of launch_v2_choice (osg_gemm.hip): no 128x128 6-stage ring; tiles 4 .. 7 (round 6, osg_gemm_wide.hip): rings of 2 / 4, 6 for 64x80
            if (tile <= 3) ok = ok && (c.nst == 2 || c.nst == 4 || (c.nst == 6 && tile >= 1) || (c.nst == 8 && tile == 2));
            else ok = ok && (c.nst == 2 || c.nst == 4 || (c.nst == 6 && tile == 6));
        } else {
            ok = c.family == 1 && (c.cfg == 0 || (c.cfg == 16 && c.splits >= 2 && c.splits <= 4)) && (c.bn == 80 || c.bn == 128 || c.bn == 160) && c.splits >= 1 && c.splits <= 64 &&
                 (c.nst == 0 || c.nst == 4 || c.nst == 8);   /* nst of a halo-convolution row = its loader waves */
        }
        k.device = 0;
        if (ok && k.M > 0 && k.N > 0 && k.K > 0 && k.batch > 0) g_table[k] = c;
    }
    fclose(f);
}
static std::set<Key> g_missed;   // DISTINCT shapes that missed (a shape that cannot be timed -- during capture, or when it consumes its own output -- misses on every launch)
bool lookup(const Key& k, Choice* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    load_locked();
    auto it = g_table.find(k);
    if (it == g_table.end()) {
        // a miss: the caller times the candidates now (and stores the winner) -- or, frozen, runs the cost model's first candidate.  Counted (osg_tune_misses)
        // and named once per shape under OSG_TUNE_LOG_MISSES=1, so that a shipped table can be completed.
        const bool first = g_missed.insert(k).second;
        static const bool log = getenv("OSG_TUNE_LOG_MISSES") != nullptr;
        if (log && first) fprintf(stderr, "[tune] miss: kind %d M %d N %d K %d batch %d H %d W %d Cin %d KW %d stride %dx%d flags %d%s\n", k.kind, k.M, k.N, k.K, k.batch, k.H, k.W, k.Cin, k.KW, k.sh,
                         k.sw, k.flags, frozen() ? " (frozen: cost-model choice, not timed)" : "");
        return false;
    }
    *out = it->second;
    return true;
}
int misses() {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_missed.size();
}
// frozen tables: remember the untimed choice of a missed shape for the rest of the process (in memory only: us = -1), so that eager launches of that shape do
// not rank the candidates again and the miss is counted once
void remember(const Key& k, const Choice& c) {
    std::lock_guard<std::mutex> lk(g_mu);
    load_locked();
    g_table.emplace(k, c);
}
void store(const Key& k, const Choice& c) {
    std::lock_guard<std::mutex> lk(g_mu);
    load_locked();
    g_table[k] = c;
    if (const char* path = cache_path()) {
        if (FILE* f = fopen(path, "a")) {
            fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %.3f\n", k.kind, k.device, k.M, k.N, k.K, k.batch, k.H, k.W, k.Cin, k.KW, k.sh,
                    k.sw, k.flags, c.family, c.cfg, c.nst, c.splits, c.bn, c.us);
            fclose(f);
        }
    }
}
}  // namespace osg_tune

extern "C" int osg_tune_misses(void) { return osg_tune::misses(); }

#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace

// ---- which XCDs does this device have (osg_common.h: xcd_ids8), and the host-mapped error flag of the in-kernel split-K fold (xcd_err) ------------------------
namespace {
__global__ void xcc_probe_kernel(int* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 15u);
}

// does an isolated launch spread its workgroups round-robin over exactly 8 XCDs whose XCC_ID are 0..7?  (true on an MI355X in SPX mode: the GroupNorm statistics
// tables then keep one copy per XCD, osg_gemm_common.h StatSink.)  Anything else -- another partition mode, fewer XCDs, a failed probe -- leaves xcd_ids8 false.
void calibrate_xcd(osg_ctx* c) {
    constexpr int kBlocks = 256;
    int* d = nullptr;
    int h[kBlocks];
    if (hipHostMalloc((void**)&c->xcd_err, 64, hipHostMallocMapped) != hipSuccess) { c->xcd_err = nullptr; return; }
    *c->xcd_err = 0;
    if (hipHostGetDevicePointer((void**)&c->xcd_err_dev, c->xcd_err, 0) != hipSuccess) return;
    if (hipMalloc((void**)&d, sizeof(h)) != hipSuccess) return;
    bool ok = true;
    for (int rep = 0; rep < 3 && ok; rep++) {
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(kBlocks), dim3(64), 0, c->compute, d);
        ok = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->compute) == hipSuccess && hipStreamSynchronize(c->compute) == hipSuccess;
        for (int i = 0; ok && i < kBlocks; i++) ok = h[i] == h[i & 7];
        for (int i = 0; ok && i < 8; i++)
            for (int j = 0; j < i; j++) ok = ok && h[i] != h[j];
    }
    (void)hipGetLastError();
    hipFree(d);
    if (getenv("OSG_XCD_DEBUG")) {
        fprintf(stderr, "[osg] XCC_ID of workgroups 0..15 of a launch:");
        for (int i = 0; i < 16; i++) fprintf(stderr, " %d", h[i]);
        fprintf(stderr, "  -> round robin over 8 XCDs: %s\n", ok ? "yes" : "no");
    }
    if (!ok) return;
    c->xcd_ids8 = true;                 // (8 distinct XCC_IDs ...
    for (int i = 0; i < 8; i++) c->xcd_ids8 = c->xcd_ids8 && h[i] >= 0 && h[i] < 8;   // ... all of them in 0..7: the per-XCD statistics tables index by XCC_ID)
}

int xcd_check(osg_ctx* c) {
    if (c->xcd_err && *(volatile int*)c->xcd_err) {
        const int what = *(volatile int*)c->xcd_err;
        *c->xcd_err = 0;
        // the last arriver that gave up left its tile's two counters as they stood (a late publication may still have bumped one): every counter back to zero, behind
        // everything the stream holds -- the captured graph stays valid (the counters are data), only this pass's results are lost
        if (c->tickets && hipMemsetAsync(c->tickets, 0, osg_ctx::kTickets * sizeof(int), c->compute) == hipSuccess) hipStreamSynchronize(c->compute);
        if (what == 2) OSG_FAIL(c, "split-K fold: a k-slice workgroup did not publish its partial sums within 20 ms (results of this pass are invalid); set OSG_SPLITK_FOLD=0");
        OSG_FAIL(c, "split-K fold: the device raised its error flag (results of this pass are invalid); set OSG_SPLITK_FOLD=0");
    }
    return 0;
}
}  // namespace

extern "C" {

void osg_range_push(const char* name) { if (roctx().push) roctx().push(name); }
void osg_range_pop(void) { if (roctx().pop) roctx().pop(); }

int osg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int osg_init(int device, osg_ctx** out) {
    if (!out) return 1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 2;  // no GPU: the product path fails loudly
    if (device < 0 || device >= n) return 3;
    if (hipSetDevice(device) != hipSuccess) return 4;
    osg_ctx* c = new osg_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->name = prop.gcnArchName;
        c->num_cu = prop.multiProcessorCount;
    }
    bool ok = hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->copy2, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_copy2, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) == hipSuccess &&
              hipEventCreate(&c->ev_t0) == hipSuccess && hipEventCreate(&c->ev_t1) == hipSuccess &&
              hipEventCreate(&c->ev_a0) == hipSuccess && hipEventCreate(&c->ev_a1) == hipSuccess;
    c->stage_bytes = 64u << 20;
    if (const char* e = getenv("OSG_COPY_STREAMS")) c->copy_streams = atoi(e) > 1 ? 2 : 1;
    for (int i = 0; ok && i < osg_ctx::kStages; i++) {
        ok = hipHostMalloc(&c->stage[i], c->stage_bytes, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&c->stage_free[i], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipMalloc((void**)&c->tickets, osg_ctx::kTickets * sizeof(int)) == hipSuccess &&
         hipMemsetAsync(c->tickets, 0, osg_ctx::kTickets * sizeof(int), c->compute) == hipSuccess &&   // on the stream the kernels run on (it does not
         hipStreamSynchronize(c->compute) == hipSuccess;                                                 // synchronise with the null stream)
    if (!ok) {
        delete c;
        return 5;
    }
    calibrate_xcd(c);
    *out = c;
    return 0;
}

void osg_destroy(osg_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (int i = 0; i < osg_ctx::kStages; i++) {
        if (c->stage[i]) hipHostFree(c->stage[i]);
        if (c->stage_free[i]) hipEventDestroy(c->stage_free[i]);
    }
    if (c->tickets) hipFree(c->tickets);
    if (c->xcd_err) hipHostFree(c->xcd_err);
    if (c->evict) hipFree(c->evict);
    if (c->ws) hipFree(c->ws);
    if (c->ws2) hipFree(c->ws2);
    if (c->ev_copy) hipEventDestroy(c->ev_copy);
    if (c->ev_copy2) hipEventDestroy(c->ev_copy2);
    if (c->copy2) hipStreamDestroy(c->copy2);
    if (c->ev_t0) hipEventDestroy(c->ev_t0);
    if (c->ev_t1) hipEventDestroy(c->ev_t1);
    if (c->ev_a0) hipEventDestroy(c->ev_a0);
    if (c->ev_a1) hipEventDestroy(c->ev_a1);
    for (auto& e : c->markers)
        if (e) hipEventDestroy(e);
    for (auto& e : c->marks)
        if (e) hipEventDestroy(e);
    if (c->compute) hipStreamDestroy(c->compute);
    if (c->copy) hipStreamDestroy(c->copy);
    delete c;
}

int osg_set_autotune(osg_ctx* c, int on) {
    if (!c) return 1;
    c->autotune = on != 0;
    return 0;
}

const char* osg_last_error(const osg_ctx* c) { return c ? c->err.c_str() : "null context"; }
const char* osg_device_name(const osg_ctx* c) { return c ? c->name.c_str() : ""; }
void* osg_stream(const osg_ctx* c) { return c ? (void*)c->compute : nullptr; }

int osg_malloc(osg_ctx* c, size_t bytes, void** dptr) {
    if (c->capturing) OSG_FAIL(c, "osg_malloc inside graph capture");
    OSG_HIP(c, hipSetDevice(c->device));
    OSG_HIP(c, hipMalloc(dptr, bytes ? bytes : 16));
    return 0;
}

int osg_free(osg_ctx* c, void* dptr) {
    if (!dptr) return 0;
    OSG_HIP(c, hipFree(dptr));
    return 0;
}

// Host->device through the pinned double-buffered ring on the COPY stream: chunk k+1 is memcpy'd into pinned memory
// while chunk k's hipMemcpyAsync is in flight; the compute stream is made to wait for the last chunk.
int osg_upload(osg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (c->capturing) OSG_FAIL(c, "osg_upload inside graph capture");
    const char* s = (const char*)src;
    char* d = (char*)dst;
    size_t off = 0;
    while (off < bytes) {
        size_t n = bytes - off < c->stage_bytes ? bytes - off : c->stage_bytes;
        int i = c->stage_next;
        c->stage_next = (i + 1) % osg_ctx::kStages;
        OSG_HIP(c, hipEventSynchronize(c->stage_free[i]));  // previous DMA out of this pinned buffer finished
        std::memcpy(c->stage[i], s + off, n);
        OSG_HIP(c, hipMemcpyAsync(d + off, c->stage[i], n, hipMemcpyHostToDevice, c->copy));
        OSG_HIP(c, hipEventRecord(c->stage_free[i], c->copy));
        off += n;
    }
    OSG_HIP(c, hipEventRecord(c->ev_copy, c->copy));
    OSG_HIP(c, hipStreamWaitEvent(c->compute, c->ev_copy, 0));
    return 0;
}

// Zero-copy variant for host memory the caller keeps alive and has page-locked with osg_host_register (a RAM weights provider's
// buffers): one hipMemcpyAsync on the COPY stream straight out of the caller's memory, the compute stream waits on its event.
int osg_upload_pinned(osg_ctx* c, void* dst, const void* pinned_src, size_t bytes) {
    if (c->capturing) OSG_FAIL(c, "osg_upload_pinned inside graph capture");
    OSG_HIP(c, hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, c->copy));
    OSG_HIP(c, hipEventRecord(c->ev_copy, c->copy));
    OSG_HIP(c, hipStreamWaitEvent(c->compute, c->ev_copy, 0));
    return 0;
}

// The streamed pass's form: enqueue only (alternating between the two H2D queues, so that one copy's completion handshake hides behind the
// other's transfer); osg_copy_fence() then makes the compute stream wait for everything enqueued so far -- once per step, not once per weight.
int osg_upload_pinned_async(osg_ctx* c, void* dst, const void* pinned_src, size_t bytes) {
    if (c->capturing) OSG_FAIL(c, "osg_upload_pinned_async inside graph capture");
    const int q = c->copy_streams > 1 ? (c->copy_rr++ & 1) : 0;
    OSG_HIP(c, hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, q ? c->copy2 : c->copy));
    c->copy_dirty[q] = true;
    return 0;
}

int osg_copy_fence(osg_ctx* c) {
    if (c->capturing) OSG_FAIL(c, "osg_copy_fence inside graph capture");
    if (c->copy_dirty[0]) {
        OSG_HIP(c, hipEventRecord(c->ev_copy, c->copy));
        OSG_HIP(c, hipStreamWaitEvent(c->compute, c->ev_copy, 0));
        c->copy_dirty[0] = false;
    }
    if (c->copy_dirty[1]) {
        OSG_HIP(c, hipEventRecord(c->ev_copy2, c->copy2));
        OSG_HIP(c, hipStreamWaitEvent(c->compute, c->ev_copy2, 0));
        c->copy_dirty[1] = false;
    }
    return 0;
}

int osg_marker_record(osg_ctx* c, int slot) {
    if (slot < 0 || slot >= osg_ctx::kMarkers) OSG_FAIL(c, "osg_marker_record: slot out of range");
    if (c->capturing) OSG_FAIL(c, "osg_marker_record inside graph capture");
    if (!c->markers[slot]) OSG_HIP(c, hipEventCreateWithFlags(&c->markers[slot], hipEventDisableTiming));
    OSG_HIP(c, hipEventRecord(c->markers[slot], c->compute));
    return 0;
}

int osg_copy_wait_marker(osg_ctx* c, int slot) {
    if (slot < 0 || slot >= osg_ctx::kMarkers) OSG_FAIL(c, "osg_copy_wait_marker: slot out of range");
    if (!c->markers[slot]) return 0;   // never recorded: nothing to wait for
    OSG_HIP(c, hipStreamWaitEvent(c->copy, c->markers[slot], 0));
    OSG_HIP(c, hipStreamWaitEvent(c->copy2, c->markers[slot], 0));
    return 0;
}

int osg_host_register(osg_ctx* c, void* ptr, size_t bytes) {
    OSG_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}

int osg_host_unregister(osg_ctx* c, void* ptr) {
    OSG_HIP(c, hipHostUnregister(ptr));
    return 0;
}

int osg_upload_sync(osg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (c->capturing) OSG_FAIL(c, "osg_upload_sync inside graph capture");
    OSG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->compute));
    OSG_HIP(c, hipStreamSynchronize(c->compute));
    return 0;
}

int osg_download(osg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (c->capturing) OSG_FAIL(c, "osg_download inside graph capture");
    OSG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->compute));
    OSG_HIP(c, hipStreamSynchronize(c->compute));
    return xcd_check(c);
}

int osg_copy(osg_ctx* c, void* dst, const void* src, size_t bytes) {
    OSG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->compute));
    return 0;
}

int osg_memset(osg_ctx* c, void* dst, int value, size_t bytes) {
    OSG_HIP(c, hipMemsetAsync(dst, value, bytes, c->compute));
    return 0;
}

int osg_sync(osg_ctx* c) {
    OSG_HIP(c, hipStreamSynchronize(c->copy));
    OSG_HIP(c, hipStreamSynchronize(c->copy2));
    OSG_HIP(c, hipStreamSynchronize(c->compute));
    return xcd_check(c);
}

int osg_graph_begin(osg_ctx* c) {
    if (c->capturing) OSG_FAIL(c, "already capturing");
    OSG_HIP(c, hipStreamSynchronize(c->copy));
    OSG_HIP(c, hipStreamSynchronize(c->copy2));
    // the arrival / departure counters (split-K tickets, GroupNorm clusters) are zero between launches by construction; a pass that ended in an error
    // may have left some behind -- a plan is captured once, right here is where they are put back (stream-ordered, outside the capture)
    if (c->tickets) OSG_HIP(c, hipMemsetAsync(c->tickets, 0, osg_ctx::kTickets * sizeof(int), c->compute));
    OSG_HIP(c, hipStreamBeginCapture(c->compute, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    return 0;
}

int osg_graph_end(osg_ctx* c, osg_graph** out) {
    if (!c->capturing) OSG_FAIL(c, "not capturing");
    c->capturing = false;
    hipGraph_t g = nullptr;
    OSG_HIP(c, hipStreamEndCapture(c->compute, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        hipGraphDestroy(g);
        OSG_FAIL(c, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    osg_graph* r = new osg_graph();
    r->graph = g;
    r->exec = ex;
    *out = r;
    return 0;
}

int osg_graph_launch(osg_ctx* c, osg_graph* g) {
    OSG_HIP(c, hipGraphLaunch(g->exec, c->compute));
    return 0;
}

void osg_graph_destroy(osg_graph* g) {
    if (!g) return;
    if (g->exec) hipGraphExecDestroy(g->exec);
    if (g->graph) hipGraphDestroy(g->graph);
    delete g;
}

int osg_timer_start(osg_ctx* c) {
    OSG_HIP(c, hipEventRecord(c->ev_t0, c->compute));
    return 0;
}

int osg_timer_mark(osg_ctx* c, int index) {
    if (index < 0 || index >= 4096) OSG_FAIL(c, "osg_timer_mark: index out of range");
    if (c->capturing) OSG_FAIL(c, "osg_timer_mark inside graph capture");
    if ((int)c->marks.size() <= index) c->marks.resize(index + 1, nullptr);
    if (!c->marks[index]) OSG_HIP(c, hipEventCreate(&c->marks[index]));
    OSG_HIP(c, hipEventRecord(c->marks[index], c->compute));
    return 0;
}

int osg_timer_between(osg_ctx* c, int a, int b, float* ms) {
    if (a < 0 || b < 0 || a >= (int)c->marks.size() || b >= (int)c->marks.size() || !c->marks[a] || !c->marks[b]) OSG_FAIL(c, "osg_timer_between: unknown mark");
    OSG_HIP(c, hipEventSynchronize(c->marks[b]));
    OSG_HIP(c, hipEventElapsedTime(ms, c->marks[a], c->marks[b]));
    return 0;
}

int osg_timer_stop(osg_ctx* c, float* ms) {
    OSG_HIP(c, hipEventRecord(c->ev_t1, c->compute));
    OSG_HIP(c, hipEventSynchronize(c->ev_t1));
    OSG_HIP(c, hipEventElapsedTime(ms, c->ev_t0, c->ev_t1));
    return 0;
}

}  // extern "C"

static int ensure_buf(osg_ctx* c, void** ptr, size_t* cur, size_t bytes);

// developer probe: OSG_KDBG=1 -> every contraction launch records per-workgroup phase timestamps into this buffer (overwritten by the next launch);
// osg_kdbg_read copies the first `bytes` of it to the host
static long long* g_kdbg = nullptr;
static long g_kdbg_wgs = 0;
long long* osg_mm::kdbg_buffer(osg_ctx* c, long workgroups) {
    static const bool on = getenv("OSG_KDBG") != nullptr;
    if (!on || c->capturing) return nullptr;
    if (workgroups > g_kdbg_wgs) {
        if (g_kdbg) hipFree(g_kdbg);
        g_kdbg_wgs = workgroups > 65536 ? workgroups : 65536;
        if (hipMalloc((void**)&g_kdbg, (size_t)g_kdbg_wgs * 64) != hipSuccess) { g_kdbg = nullptr; g_kdbg_wgs = 0; return nullptr; }
    }
    return g_kdbg;
}
extern "C" int osg_kdbg_read(osg_ctx* c, void* host, size_t bytes) {
    if (!g_kdbg || bytes > (size_t)g_kdbg_wgs * 64) OSG_FAIL(c, "osg_kdbg_read: no probe buffer (set OSG_KDBG=1)");
    OSG_HIP(c, hipStreamSynchronize(c->compute));
    OSG_HIP(c, hipMemcpy(host, g_kdbg, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int osg_ensure_workspace(osg_ctx* c, size_t bytes) { return ensure_buf(c, &c->ws, &c->ws_bytes, bytes); }
int osg_ensure_workspace2(osg_ctx* c, size_t bytes) { return ensure_buf(c, &c->ws2, &c->ws2_bytes, bytes); }

static int ensure_buf(osg_ctx* c, void** ptr, size_t* cur, size_t bytes) {
    if (bytes <= *cur) return 0;
    if (c->capturing) OSG_FAIL(c, "workspace growth inside graph capture (run the pass once eagerly first)");
    OSG_HIP(c, hipStreamSynchronize(c->compute));
    if (*ptr) OSG_HIP(c, hipFree(*ptr));
    *ptr = nullptr;
    *cur = 0;
    size_t n = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    OSG_HIP(c, hipMalloc(ptr, n));
    *cur = n;
    return 0;
}
