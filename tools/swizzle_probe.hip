// Dev probe (GPU box), round 6: why does the k loop of gemm2_kernel receive ~22-25 B/clk per CU (hot operands or cold, every tile, every ring depth, every row pitch:
// profiles/r06_gemm_kloop_pitch_probe*.txt) when tools/fillrate_probe.hip measures 52-62 B/clk per CU for the same `buffer_load_dwordx4 ... lds` instruction on the same
// 8-rows-x-128-B pattern?  What the GEMM does and that probe does not:
//   SWZ  1: the 16-byte chunks of a 128-byte row piece are requested in XOR-PERMUTED lane order (chunk (lane & 7) ^ (lane >> 3): the bank swizzle of the LDS image is
//           applied on the source address); 2: rotated order ((lane & 7) + (lane >> 3)) & 7; 0: ascending (the fill-rate probe)
//   RING 1: requests trickle -- per k-tile step every wave waits for its oldest tile (counted vmcnt), meets the others at a barrier and issues 4 more -- instead of
//           16 at once and a full drain
//   READS 1: between the barrier and the next step every wave reads 8 fragments (ds_read_b128) out of the landing zone, as the math does
//   build: hipcc --offload-arch=gfx950 -O3 tools/swizzle_probe.hip -o tools/_build/swizzle_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int kSteps = 1600;   // k-tile steps of 16 KiB per workgroup (4 wave-loads per wave)

template <int SWZ, int RING, int READS>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, float* __restrict__ sink, int S, int panels) {
    extern __shared__ char smem[];   // 4 stages x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = buf + (size_t)(blockIdx.x % panels) * 256 * S;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 256 * S, 0x00020000);
    const int kts = S / 128;
    int kt = blockIdx.x % kts;
    const int rsub = lane >> 3;
    const int ch = SWZ == 1 ? ((lane & 7) ^ rsub) : SWZ == 2 ? (((lane & 7) + rsub) & 7) : (lane & 7);
    // a step = one 128-row x 128-B k-tile (16 wave-loads, 4 per wave); the 128 rows of step s: rows (s & 1) * 128 .. of the 256-row panel
    unsigned off[4];
#pragma unroll
    for (int j = 0; j < 4; j++) off[j] = (unsigned)(((j * 4 + wave) * 8 + rsub) * S + ch * 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    auto issue = [&](int stage, int step) {
        const unsigned soff = (unsigned)((step & 1) * 128 * S + kt * 128);
#pragma unroll
        for (int j = 0; j < 4; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + stage * 16384 + (j * 4 + wave) * 1024), 16, off[j], soff, 0, 0);
        if (step & 1) kt = kt + 1 < kts ? kt + 1 : 0;
    };
    if (RING) {
        issue(0, 0); issue(1, 1); issue(2, 2);
        int cur = 0, nxt = 3;
        for (int s = 0; s < kSteps; s++) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue(nxt, s + 3);
            if (READS) {
                const char* St = smem + cur * 16384;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int row = (wave & 1) * 32 + (r & 1) * 16 + ((r >> 1) & 1) * 64 + frow;
                    acc += *reinterpret_cast<const f32x4*>(St + ((row * 128 + ((fq ^ (row & 7)) << 4)) ^ ((r >> 2) << 6)));
                }
            }
            cur = (cur + 1) & 3;
            nxt = (nxt + 1) & 3;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int s = 0; s < kSteps; s += 4) {
            issue(0, s); issue(1, s + 1); issue(2, s + 2); issue(3, s + 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[tid] = acc[0] + smem[tid];
}

// What shares what: per step every wave issues ND LDS-DMA wave-loads (1 KiB each, the 8-row pattern), NV 1-KiB register loads (contiguous: a fragment-major copy),
// and NR ds_read_b128 of the landing zone; ring of 4 steps, counted vmcnt, one barrier per step (register loads by inline asm: hipcc's own vmcnt bookkeeping drains the
// queue at the loop's back edge when a register ring is written by ordinary loads).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int ND, int NV, int NR, int NM = 0>
__device__ __forceinline__ void kmix_body(const char* __restrict__ buf, float* __restrict__ sink, int S, int panels) {
    extern __shared__ char smem[];   // 4 stages x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = buf + (size_t)(blockIdx.x % panels) * 256 * S;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 256 * S, 0x00020000);
    const int kts = S / 128;
    int kt = blockIdx.x % kts;
    const int rsub = lane >> 3;
    unsigned off[ND ? ND : 1];
#pragma unroll
    for (int j = 0; j < ND; j++) off[j] = (unsigned)(((j * 4 + wave) * 8 + rsub) * S + ((lane & 7) ^ rsub) * 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 breg[4][NV ? NV : 1];
    f32x4 macc[NM ? NM / 2 : 1];
#pragma unroll
    for (int m = 0; m < (NM ? NM / 2 : 1); m++) macc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    int bpos = __builtin_amdgcn_readfirstlane(((blockIdx.x * 4 + wave) * 4096) % 65536);   // this wave's contiguous stream inside the panel (wraps)
    const int bwrap = 256 * S - 4096 * NV;
    auto issue = [&](int stage, auto st) {
        constexpr int ST = decltype(st)::value;
        const unsigned soff = (unsigned)(kt * 128);
#pragma unroll
        for (int j = 0; j < ND; j++) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + stage * 16384 + (j * 4 + wave) * 1024), 16, off[j], soff, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; j++) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(breg[ST][j]) : "v"(lane * 16 + j * 1024), "s"(rs), "s"(bpos) : "memory");
        kt = kt + 1 < kts ? kt + 1 : 0;
        bpos = bpos + 65536 <= bwrap ? bpos + 65536 : bpos % 65536;
    };
    using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>; using c2 = std::integral_constant<int, 2>; using c3 = std::integral_constant<int, 3>;
    issue(0, c0{}); issue(1, c1{}); issue(2, c2{});
    auto step = [&](auto cur_, auto nxt_) {
        constexpr int CUR = decltype(cur_)::value, NXT = decltype(nxt_)::value;
        if (ND + NV) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (ND + NV)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(NXT, nxt_);
        const char* St = smem + CUR * 16384;
        if constexpr (NM == 0) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int row = (r & 7) * 16 + frow;
                acc += *reinterpret_cast<const f32x4*>(St + ((row * 128 + ((fq ^ (row & 7)) << 4)) ^ ((r >> 3) << 6)));
            }
        } else {
            // NM MFMAs per step on the NR fragments (half of them "A", half "B"; accumulators: NM / 2 independent ones, each used twice, as the two k-halves of a tile)
            f16x8 fr[NR];
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int row = (r & 7) * 16 + frow;
                fr[r] = *reinterpret_cast<const f16x8*>(St + ((row * 128 + ((fq ^ (row & 7)) << 4)) ^ ((r >> 3) << 6)));
            }
#pragma unroll
            for (int m = 0; m < NM; m++) macc[m % (NM / 2)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[m % (NR / 2)], fr[NR / 2 + (m / (NR / 2)) % (NR / 2)], macc[m % (NM / 2)], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NV; j++) acc += breg[CUR][j];
    };
    for (int s = 0; s < kSteps; s += 4) { step(c0{}, c3{}); step(c1{}, c0{}); step(c2{}, c1{}); step(c3{}, c2{}); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < (NM ? NM / 2 : 1); m++) acc += macc[m];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[tid] = acc[0] + smem[tid];
}

// (hipcc emits no host stub for a __global__ template whose body holds a generic lambda)
template <int ND, int NV, int NR, int NM = 0>
__global__ __launch_bounds__(256) void kmix(const char* __restrict__ buf, float* __restrict__ sink, int S, int panels) { kmix_body<ND, NV, NR, NM>(buf, sink, S, panels); }

template <int ND, int NV, int NR, int NM = 0>
static void run_mix(const char* d, float* sink, int cus, double ghz, int S) {
    auto kern = kmix<ND, NV, NR, NM>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 65536, 0, d, sink, S, 8);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double clk = best * 1e-3 * ghz * 1e9 / kSteps;
    printf("per step and workgroup: %2d KiB by LDS-DMA, %2d KiB into registers, %3d KiB of fragment reads (%2d per wave), %2d MFMAs per wave: %5.0f clk per step  = %5.1f B/clk per CU loaded, %5.1f B/clk read\n", ND * 4, NV * 4,
           NR * 4, NR, NM, clk, (ND + NV) * 4096.0 / clk, NR * 4096.0 / clk);
}

template <int SWZ, int RING, int READS>
static void run(const char* d, float* sink, int blocks, int cus, double ghz, int S, int panels) {
    auto kern = k<SWZ, RING, READS>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, d, sink, S, panels);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, d, sink, S, panels);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * kSteps * 16384.0;
    const double active = blocks < cus ? blocks : cus;
    printf("order %-9s %-22s %-9s row pitch %5d B, %d panel(s), blocks %4d: %7.3f ms  %6.2f TB/s  %5.1f B/clk per active CU  (%4.0f clk per 16-KiB step)\n",
           SWZ == 1 ? "xor" : SWZ == 2 ? "rotated" : "ascending", RING ? "ring of 4, barrier/step" : "16 at once, full drain", READS ? "+8 reads" : "", S, panels, blocks, best,
           bytes / best * 1e-9, bytes / (best * 1e-3) / (ghz * 1e9) / active, best * 1e-3 * ghz * 1e9 / kSteps / (blocks > cus ? (double)blocks / cus : 1.0));
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz\n", prop.gcnArchName, cus, ghz);
    char* d;
    float* sink;
    hipMalloc((void**)&d, (size_t)8 * 256 * 5120);
    hipMemset(d, 1, (size_t)8 * 256 * 5120);
    hipMalloc((void**)&sink, 4096);
    for (int S : {1280, 2560})
        for (int panels : {8, 1}) {
            run<0, 0, 0>(d, sink, cus, cus, ghz, S, panels);
            run<1, 0, 0>(d, sink, cus, cus, ghz, S, panels);
            run<2, 0, 0>(d, sink, cus, cus, ghz, S, panels);
            run<0, 1, 0>(d, sink, cus, cus, ghz, S, panels);
            run<1, 1, 0>(d, sink, cus, cus, ghz, S, panels);
            run<0, 1, 1>(d, sink, cus, cus, ghz, S, panels);
            run<1, 1, 1>(d, sink, cus, cus, ghz, S, panels);
        }
    run_mix<4, 0, 0>(d, sink, cus, ghz, 2560);
    run_mix<4, 0, 8>(d, sink, cus, ghz, 2560);
    run_mix<4, 0, 16>(d, sink, cus, ghz, 2560);
    run_mix<0, 0, 8>(d, sink, cus, ghz, 2560);
    run_mix<0, 0, 16>(d, sink, cus, ghz, 2560);
    run_mix<0, 4, 0>(d, sink, cus, ghz, 2560);
    run_mix<0, 4, 8>(d, sink, cus, ghz, 2560);
    run_mix<2, 4, 0>(d, sink, cus, ghz, 2560);
    run_mix<2, 4, 8>(d, sink, cus, ghz, 2560);
    run_mix<2, 2, 8>(d, sink, cus, ghz, 2560);
    run_mix<2, 0, 8>(d, sink, cus, ghz, 2560);
    run_mix<2, 0, 16>(d, sink, cus, ghz, 2560);
    run_mix<8, 0, 16>(d, sink, cus, ghz, 2560);
    run_mix<4, 0, 8, 8>(d, sink, cus, ghz, 2560);     // the 64 x 64 tile's step
    run_mix<0, 0, 8, 8>(d, sink, cus, ghz, 2560);
    run_mix<8, 0, 16, 32>(d, sink, cus, ghz, 2560);   // the 128 x 128 tile's step
    run_mix<0, 0, 16, 32>(d, sink, cus, ghz, 2560);
    run_mix<8, 0, 0, 0>(d, sink, cus, ghz, 2560);
    return 0;
}
