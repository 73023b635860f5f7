// Dev probe (GPU box), round 6: what does the fused epilogue's OUTPUT cost as the MFMA accumulator layout writes it -- 8 bytes per lane, a wave-instruction = 16 rows x 32
// bytes -- against the same tile transposed through LDS first and written as whole rows (16 bytes per lane, a wave-instruction = 8 rows x 128 contiguous bytes)?
// tools/gemm_kloop_probe.py shows 2.7-4.4 us of epilogue on the 128-row tiles of gemm2_kernel for 32-40 KiB of output per workgroup.
//   build: hipcc --offload-arch=gfx950 -O3 tools/epi_store_probe.hip -o tools/_build/epi_store_probe      run: tools/_build/epi_store_probe
// One workgroup (256 threads, 4 waves as 2 x 2) per BM x BN tile of an [M][N] f16 matrix; values come from registers (no loads); NBUF output matrices are
// cycled so that a launch does not find its lines dirty in the L2 from the launch before.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: accumulator layout, 8-byte buffer stores (gemm_epilogue_fast today)    1: through LDS, 16-byte row stores    2: no stores at all (launch floor)
template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void store_kernel(f16* __restrict__ C, int M, int N, float seed) {
    constexpr int TM = BM / 32, TN = BN / 32;                 // 16 x 16 blocks per wave (2 x 2 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    f16x4 o[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[i][j][r] = (f16)(seed + (float)(i * 7 + j * 3 + r + lane));
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, 0x80000000u, 0x00020000);
    if constexpr (MODE == 0) {
        const int nb = n0 + wn0 + (lane >> 4) * 4, mb = m0 + wm0 + (lane & 15);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const unsigned ro = (unsigned)(((long)(mb + i * 16) * N + nb) * 2);
#pragma unroll
            for (int j = 0; j < TN; j++) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o[i][j]), rsC, ro + j * 32, 0, 0);
        }
    } else if constexpr (MODE == 1) {
        // the wave's (BM/2) x (BN/2) sub-tile, row-major in its own LDS region, pitch = row bytes + 16 (rows 4 banks apart)
        constexpr int RB = BN / 2 * 2, PITCH = RB + 16, CH = RB / 16;     // bytes per row, chunks of 16 bytes per row
        char* st = smem + wave * (BM / 2) * PITCH;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                *reinterpret_cast<f16x4*>(st + (i * 16 + (lane & 15)) * PITCH + (j * 16 + (lane >> 4) * 4) * 2) = o[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (one wave: its own LDS writes are visible to its lanes once they have completed)
        constexpr int TOT = (BM / 2) * CH;
#pragma unroll
        for (int it = 0; it < (TOT + 63) / 64; it++) {
            const int idx = it * 64 + lane, row = idx / CH, ch = idx % CH;
            if (TOT % 64 == 0 || idx < TOT) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * PITCH + ch * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsC, (unsigned)(((long)(m0 + wm0 + row) * N + n0 + wn0) * 2 + ch * 16), 0, 0);
            }
        }
    } else {
        if (seed == 12345.f) C[tid] = o[0][0][0];
    }
}

template <int BM, int BN, int MODE>
static float run(f16** bufs, int nbuf, int M, int N, int reps) {
    auto k = store_kernel<BM, BN, MODE>;
    const int lds = MODE == 1 ? 4 * (BM / 2) * (BN + 16) : 0;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    const int grid = (M / BM) * (N / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int r = 0; r < reps + 3; r++) {
        CK(hipEventRecord(e0, 0));
        for (int b = 0; b < nbuf; b++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, bufs[b], M, N, (float)r);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) t.push_back(ms * 1e3f / nbuf);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <int BM, int BN>
static void shape(int M, int N, f16** bufs, int nbuf) {
    const float a = run<BM, BN, 0>(bufs, nbuf, M, N, 15), b = run<BM, BN, 1>(bufs, nbuf, M, N, 15), c = run<BM, BN, 2>(bufs, nbuf, M, N, 15);
    const float a2 = run<BM, BN, 0>(bufs, nbuf, M, N, 15), b2 = run<BM, BN, 1>(bufs, nbuf, M, N, 15);
    const double mb = (double)M * N * 2 / 1e6;
    printf("[%5d x %5d] tile %3d x %3d  %4d workgroups  %6.2f MB | 8-byte accumulator layout %6.2f / %6.2f us (%5.2f TB/s)  rows through LDS %6.2f / %6.2f us (%5.2f TB/s)  no stores %5.2f us\n", M, N, BM, BN,
           (M / BM) * (N / BN), mb, a, a2, mb / std::min(a, a2) * 1e-6 * 1e6, b, b2, mb / std::min(b, b2) * 1e-6 * 1e6, c);
}

int main() {
    const int nbuf = 24;                       // 24 x up to 21 MB: a launch's lines are long gone from the L2s (32 MB) and mostly from the MALL when it comes round again
    std::vector<f16*> bufs(nbuf);
    for (auto& p : bufs) CK(hipMalloc((void**)&p, (size_t)8192 * 1280 * 2));
    printf("# back-to-back launches cycling over %d output matrices (dependent on one stream), median of 15 x %d launches; TB/s = output bytes / launch time\n", nbuf, nbuf);
    shape<128, 128>(8192, 1280, bufs.data(), nbuf);
    shape<128, 128>(8192, 640, bufs.data(), nbuf);
    shape<128, 160>(8192, 960, bufs.data(), nbuf);
    shape<128, 160>(8192, 320, bufs.data(), nbuf);
    shape<64, 64>(8192, 320, bufs.data(), nbuf);
    shape<64, 64>(2048, 640, bufs.data(), nbuf);
    shape<128, 128>(2048, 2560, bufs.data(), nbuf);
    shape<128, 160>(2048, 1920, bufs.data(), nbuf);
    shape<64, 64>(2048, 1280, bufs.data(), nbuf);
    shape<64, 64>(512, 1280, bufs.data(), nbuf);
    shape<128, 128>(512, 5120, bufs.data(), nbuf);
    return 0;
}
