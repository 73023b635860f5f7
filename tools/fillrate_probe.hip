// Dev probe (GPU box): how many bytes per clock can ONE CU pull out of the L2 -- through the LDS-DMA path (buffer_load ... lds, what gemm2_kernel /
// conv3x3_kernel stage their operands with), through ordinary loads into VGPRs, and through both at once?  The contraction kernels of the UNet sit at
// ~23 B/clk/CU of operand fill; if register loads ride a separate budget, a kernel that keeps one operand out of the LDS can go past it.
//   build: hipcc --offload-arch=gfx950 -O3 tools/fillrate_probe.hip -o tools/_build/fillrate_probe      run: tools/_build/fillrate_probe
// Every workgroup (256 threads) streams its own 64 KiB window of a buffer over and over (L2-resident after the first sweep; the `shared` variants make
// the four waves of a workgroup read the SAME 16 KiB -- the B-fragment case -- so three of four requests can hit the L1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kWindow = 64 * 1024;   // bytes per workgroup
constexpr int kIters = 400;          // sweeps of the window

// MODE 0: LDS-DMA only   1: VGPR only   2: both, alternating   3: VGPR only, waves share addresses   4: LDS-DMA, waves share addresses
template <int MODE>
__global__ __launch_bounds__(256) void fill_kernel(const char* __restrict__ buf, float* __restrict__ sink, int windows) {
    extern __shared__ char smem[];   // 32 KiB landing zone (one wave-instruction = 1 KiB; up to 4 workgroups per CU)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = buf + (size_t)(blockIdx.x % windows) * kWindow;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, kWindow, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr bool SHARED = MODE == 3 || MODE == 4;
    // a sweep: every wave issues 16 x 16 B per lane = 16 KiB (4 waves: the whole 64 KiB window; SHARED: all four read the first 16 KiB)
    for (int it = 0; it < kIters; it++) {
        f32x4 r[16];
        unsigned soff = 0;
        asm volatile("" : "+s"(soff));   // (opaque to the optimiser: the loads stay inside the loop)
#pragma unroll
        for (int j = 0; j < 16; j++) {   // all 16 requests of the sweep are issued before the first result is touched
            const unsigned off = (unsigned)(((SHARED ? 0 : wave) * 16 + j) * 1024 + lane * 16);
            const bool dma = MODE == 0 || MODE == 4 || (MODE == 2 && (j & 1) == 0);
            if (dma) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * 8 + (j & 7)) * 1024), 16, off, soff, 0, 0);
            else r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const bool dma = MODE == 0 || MODE == 4 || (MODE == 2 && (j & 1) == 0);
            if (!dma) acc += r[j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[tid] = acc[0] + smem[tid];
}

// The GEMM's pattern: a wave-instruction fetches 8 ROWS x 128 B (row stride S bytes, the k extent of a row-major operand), a workgroup 256 rows per
// k-tile, the next k-tile sits 128 B further along every row.  All workgroups read ONE 256-row panel (the shared-operand case; it stays in the L2).
// DMA = true: buffer_load ... lds, false: into VGPRs.
template <bool DMA>
__global__ __launch_bounds__(256) void fill_rows_kernel(const char* __restrict__ buf, float* __restrict__ sink, int S, int panels) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = buf + (size_t)(blockIdx.x % panels) * 256 * S;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 256 * S, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kts = S / 128;
    int kt = blockIdx.x % kts;   // (workgroups start at different k-tiles, as tiles of different layers' progress would)
    for (int it = 0; it < kIters; it++) {
        f32x4 r[16];
        unsigned soff = 0;
        asm volatile("" : "+s"(soff));
#pragma unroll
        for (int j = 0; j < 16; j++) {   // two k-tiles per sweep: 2 x 8 instructions per wave = 64 KiB per workgroup
            const int k2 = kt + (j >> 3) < kts ? kt + (j >> 3) : 0;
            const unsigned off = (unsigned)((wave * 64 + (j & 7) * 8 + (lane >> 3)) * S + k2 * 128 + (lane & 7) * 16);
            if (DMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * 8 + (j & 7)) * 1024), 16, off, soff, 0, 0);
            else r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
        }
        if (!DMA) {
#pragma unroll
            for (int j = 0; j < 16; j++) acc += r[j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        kt = kt + 2 < kts ? kt + 2 : 0;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[tid] = acc[0] + smem[tid];
}

template <bool DMA>
static void run_rows(const char* d, float* sink, int blocks, int cus, double ghz, int S, int panels, bool lockstep) {
    auto k = fill_rows_kernel<DMA>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 32768, 0, d, sink, S, panels);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 32768, 0, d, sink, S, panels);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * kIters * (double)kWindow;
    const double active = blocks < cus ? blocks : cus;
    printf("%-8s 8 rows x 128 B per instruction, row stride %6d B, %d panel(s) of 256 rows  blocks %4d: %7.3f ms  %7.2f TB/s  %6.1f B/clk per active CU\n", DMA ? "LDS-DMA" : "VGPR", S,
           panels, blocks, ms, bytes / ms * 1e-9, bytes / (ms * 1e-3) / (ghz * 1e9) / active);
    (void)lockstep;
}

template <int MODE>
static void run(const char* what, const char* d, float* sink, int blocks, int cus, double ghz) {
    auto k = fill_kernel<MODE>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int windows = 256;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 32768, 0, d, sink, windows);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 32768, 0, d, sink, windows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const bool shared = MODE == 3 || MODE == 4;
    const double bytes = (double)blocks * kIters * (double)kWindow;   // requested bytes (SHARED: 4 x 16 KiB of the same lines)
    const double active = blocks < cus ? blocks : cus;
    printf("%-44s blocks %4d: %7.3f ms  %7.2f TB/s requested  %6.1f B/clk per active CU%s\n", what, blocks, ms, bytes / ms * 1e-9, bytes / (ms * 1e-3) / (ghz * 1e9) / active,
           shared ? "  (4 waves x the same 16 KiB)" : "");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz (B/clk figures use this clock)\n", prop.gcnArchName, cus, ghz);
    char* d;
    float* sink;
    hipMalloc((void**)&d, 256 * kWindow);
    hipMemset(d, 1, 256 * kWindow);
    hipMalloc((void**)&sink, 4096);
    for (int blocks : {cus / 2, cus, 2 * cus, 4 * cus}) {
        run<0>("LDS-DMA (buffer_load ... lds)", d, sink, blocks, cus, ghz);
        run<1>("VGPR (buffer_load_dwordx4)", d, sink, blocks, cus, ghz);
        run<2>("both, alternating", d, sink, blocks, cus, ghz);
        run<3>("VGPR, the 4 waves share their addresses", d, sink, blocks, cus, ghz);
        run<4>("LDS-DMA, the 4 waves share their addresses", d, sink, blocks, cus, ghz);
    }
    printf("\n");
    char* d2;
    hipMalloc((void**)&d2, (size_t)8 * 256 * 20480);
    hipMemset(d2, 1, (size_t)8 * 256 * 20480);
    for (int blocks : {cus, 2 * cus})
        for (int S : {640, 2560, 5120, 10240, 10240 + 128, 20480}) {
            run_rows<true>(d2, sink, blocks, cus, ghz, S, 1, false);
            run_rows<false>(d2, sink, blocks, cus, ghz, S, 1, false);
            run_rows<true>(d2, sink, blocks, cus, ghz, S, 8, false);
        }
    return 0;
}
