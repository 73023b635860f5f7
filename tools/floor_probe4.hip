// Dev probe (GPU box), round 6: WHY does a trivial node cost ~4.7 us inside the captured UNet pass and ~1.7 us in a chain of trivial kernels?
// Hypothesis: inside the pass every kernel's CODE, kernel-argument block and first data lines were last touched one pass (1.7 GB of weights) ago -- out of the 8 L2s
// and of the 256 MB Infinity Cache -- so a dispatch is a chain of dependent COLD misses (instruction fetch -> s_load of the kernargs -> first data load), each ~1 us,
// where the chain of trivial kernels hits L2 every time.  Controlled experiment: graphs of n DISTINCT kernels, each preceded by a "polluter" that streams 384 MB
// (> the Infinity Cache) so that everything the next kernel needs is cold; the same graphs with every kernel run twice (the second run is hot); kernels that read no
// kernel arguments; kernels that execute ~1500 straight-line instructions (a real kernel's prologue + epilogue); and the polluter PREFETCHING the next kernel's code
// (its address learned from s_getpc in a warm-up run) into every XCD's L2 while it finishes.
//   build: hipcc --offload-arch=gfx950 -O3 tools/floor_probe4.hip -o tools/_build/floor_probe4      run: tools/_build/floor_probe4 [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NK = 32;                       // distinct kernels per family
__device__ unsigned long long g_pc[3][NK];   // code address of every kernel (s_getpc near its entry), written on every run
__device__ float g_out[3][NK];

__device__ __forceinline__ unsigned long long get_pc() {
    unsigned long long pc;
    asm volatile("s_getpc_b64 %0" : "=s"(pc));
    return pc;
}

// family 0: no kernel arguments at all (the store address comes from the code: s_getpc + relocation)
template <int ID>
__global__ void k_noarg() {
    const unsigned long long pc = get_pc();
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_pc[0][ID] = pc; g_out[0][ID] = (float)ID; }
}
// family 1: one pointer argument, read-modify-write of one float behind it
template <int ID>
__global__ void k_arg(float* p) {
    const unsigned long long pc = get_pc();
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_pc[1][ID] = pc; p[ID * 64] += 1.f; }
}
// family 2: ... and ~1500 straight-line instructions executed once by every wave (24 x 64-instruction blocks; ~12 KB of code)
template <int ID>
__global__ void k_fat(float* p, int seed) {
    const unsigned long long pc = get_pc();
    unsigned v = threadIdx.x + seed, s = seed | 1;
#pragma unroll
    for (int i = 0; i < 24; i++) {
        asm volatile(
            ".rept 32\n v_mad_u32_u24 %0, %0, %1, %0\n s_add_u32 %1, %1, 2\n .endr\n"
            : "+v"(v), "+s"(s) : : "scc");
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_pc[2][ID] = pc; p[ID * 64] += (float)(v & 1); }
}

// streams n16 float4 (grid-stride); pf != 0: every workgroup's first wave then requests pf_bytes at pf (the NEXT kernel's code) without waiting for them
template <bool NT>
__global__ __launch_bounds__(256) void k_pollute(const float4* __restrict__ buf, size_t n16, float* sink, const char* pf, int pf_bytes) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4v*>(buf) + i) : reinterpret_cast<const f4v*>(buf)[i];   // NT: global_load_dwordx4 ... nt (streaming: does it spare the Infinity Cache?)
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.789f) sink[0] = acc;
    if (pf && threadIdx.x < 64) {
        for (int o = threadIdx.x * 16; o < pf_bytes; o += 1024) {
            float4 t;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(pf + o) : "memory");
        }
    }
}
// keeps the families' code away from the end of the loaded segment (the prefetch reads a few KB past a kernel's entry)
__global__ void k_tail_guard(float* p) {
    unsigned v = threadIdx.x, s = 1;
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile(".rept 32\n v_mad_u32_u24 %0, %0, %1, %0\n s_add_u32 %1, %1, 2\n .endr\n" : "+v"(v), "+s"(s) : : "scc");
    if (v == 0xdeadbeef) p[0] = 1.f;
}

typedef void (*Launch)(hipStream_t, float*, int grid);
template <int F, int ID> static void launch_one(hipStream_t st, float* p, int grid) {
    if constexpr (F == 0) hipLaunchKernelGGL(k_noarg<ID>, dim3(grid), dim3(256), 0, st);
    else if constexpr (F == 1) hipLaunchKernelGGL(k_arg<ID>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_fat<ID>, dim3(grid), dim3(256), 0, st, p, 3);
}
template <int F, int... I> static void fill(std::vector<Launch>& v, std::integer_sequence<int, I...>) { (v.push_back(&launch_one<F, I>), ...); }

struct Graph { hipGraphExec_t ex; const char* what; int nodes_of_interest; std::vector<float> us; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 9;
    const size_t pol_bytes = (size_t)384 << 20;
    float4* big; CK(hipMalloc((void**)&big, pol_bytes)); CK(hipMemset(big, 0, pol_bytes));
    float* d; CK(hipMalloc((void**)&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<Launch> fam[3];
    fill<0>(fam[0], std::make_integer_sequence<int, NK>{});
    fill<1>(fam[1], std::make_integer_sequence<int, NK>{});
    fill<2>(fam[2], std::make_integer_sequence<int, NK>{});
    hipLaunchKernelGGL(k_tail_guard, dim3(1), dim3(64), 0, st, d);
    for (int f = 0; f < 3; f++) for (auto l : fam[f]) l(st, d, 256);      // warm-up: loads the code objects, fills g_pc
    CK(hipStreamSynchronize(st));
    unsigned long long pc[3][NK];
    CK(hipMemcpyFromSymbol(pc, HIP_SYMBOL(g_pc), sizeof(pc)));
    printf("code addresses: noarg[0] %llx noarg[1] %llx (stride %lld B)  arg[0] %llx  fat[0] %llx fat[1] %llx (stride %lld B)\n", pc[0][0], pc[0][1], (long long)(pc[0][1] - pc[0][0]),
           pc[1][0], pc[2][0], pc[2][1], (long long)(pc[2][1] - pc[2][0]));
    auto pollute = [&](const char* pf, int pf_bytes) { hipLaunchKernelGGL(k_pollute<false>, dim3(2048), dim3(256), 0, st, big, pol_bytes / 16, d + 4096, pf, pf_bytes); };
    auto pollute_nt = [&]() { hipLaunchKernelGGL(k_pollute<true>, dim3(2048), dim3(256), 0, st, big, pol_bytes / 16, d + 4096, (const char*)nullptr, 0); };
    auto code_of = [&](int f, int i, int back) { return (const char*)((pc[f][i] - back) & ~255ull); };

    std::vector<Graph> graphs;
    auto capture = [&](const char* what, int noi, std::function<void()> body) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        body();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
        graphs.push_back(Graph{ex, what, noi, {}});
    };
    const int grids[2] = {1, 256};
    static char names[96][160];
    int nn = 0;
    capture("[P] x n  (polluter alone)", NK, [&] { for (int i = 0; i < NK; i++) pollute(nullptr, 0); });
    capture("[Pnt] x n  (polluter with nt loads alone)", NK, [&] { for (int i = 0; i < NK; i++) pollute_nt(); });
    for (int gi = 0; gi < 2; gi++) {
        const int grid = grids[gi];
        for (int f = 0; f < 3; f++) {
            const char* fn = f == 0 ? "no-arg" : f == 1 ? "1-arg" : "fat (~1500 instr)";
            snprintf(names[nn], 160, "%-18s grid %3d: [K_i] x n          (distinct kernels, hot)", fn, grid);
            capture(names[nn++], NK, [&] { for (int i = 0; i < NK; i++) fam[f][i](st, d, grid); });
            snprintf(names[nn], 160, "%-18s grid %3d: [P, K_i] x n       (cold)", fn, grid);
            capture(names[nn++], NK, [&] { for (int i = 0; i < NK; i++) { pollute(nullptr, 0); fam[f][i](st, d, grid); } });
            snprintf(names[nn], 160, "%-18s grid %3d: [P, K_i, K_i] x n  (cold, then hot)", fn, grid);
            capture(names[nn++], NK, [&] { for (int i = 0; i < NK; i++) { pollute(nullptr, 0); fam[f][i](st, d, grid); fam[f][i](st, d, grid); } });
            snprintf(names[nn], 160, "%-18s grid %3d: [Pnt, K_i] x n     (polluter streams with nt loads)", fn, grid);
            capture(names[nn++], -1, [&] { for (int i = 0; i < NK; i++) { pollute_nt(); fam[f][i](st, d, grid); } });
            const int pfb = f == 2 ? 14 * 1024 : 1024;
            snprintf(names[nn], 160, "%-18s grid %3d: [P+prefetch(code of K_i, %d B), K_i] x n", fn, grid, pfb);
            capture(names[nn++], NK, [&] { for (int i = 0; i < NK; i++) { pollute(code_of(f, i, 256), pfb); fam[f][i](st, d, grid); } });
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& g : graphs) { CK(hipGraphLaunch(g.ex, st)); }
    CK(hipStreamSynchronize(st));
    for (int r = 0; r < rounds; r++)
        for (auto& g : graphs) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(g.ex, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            g.us.push_back(ms * 1e3f);
        }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const float tp = med(graphs[0].us);
    printf("%-90s %9.2f us per replay = %7.2f us per polluter (384 MB: %.2f TB/s)\n", graphs[0].what, tp, tp / NK, 384.0 * 1.048576e6 / (tp / NK * 1e-6) / 1e12);
    const float tpn = med(graphs[1].us);
    printf("%-90s %9.2f us per replay = %7.2f us per polluter\n", graphs[1].what, tpn, tpn / NK);
    for (size_t k = 2; k < graphs.size(); k++) {
        const float t = med(graphs[k].us);
        if (graphs[k].nodes_of_interest < 0) { printf("%-90s %9.2f us per replay -> %6.2f us per K node on top of the nt polluters\n", graphs[k].what, t, (t - tpn) / NK); continue; }
        const bool has_p = strstr(graphs[k].what, "[P") != nullptr, twice = strstr(graphs[k].what, "K_i, K_i") != nullptr;
        if (!has_p) printf("%-90s %9.2f us per replay -> %6.2f us per node\n", graphs[k].what, t, t / NK);
        else if (!twice) printf("%-90s %9.2f us per replay -> %6.2f us per K node on top of the polluters\n", graphs[k].what, t, (t - tp) / NK);
        else {
            const float cold = med(graphs[k - 1].us);
            printf("%-90s %9.2f us per replay -> the second (hot) K node costs %6.2f us\n", graphs[k].what, t, (t - cold) / NK);
        }
    }
    return 0;
}
