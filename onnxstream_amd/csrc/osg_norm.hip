// libosgpu: normalisation / row reductions (HBM-bound; wavefront shuffle + LDS block reductions, f32 math).
//   osg_instance_norm   <- Model::run InstanceNormalization (reference onnxstream.cpp:4788-5055): mean, then variance
//                          of the deviations, y = scale*(x-mean)/sqrt(var+eps)+bias in f32, one rounding.
//   osg_group_norm_nhwc <- the exported GroupNorm pattern Reshape/InstanceNorm/Reshape/Mul/Add(+Sigmoid,Mul) fused, NHWC.
//   osg_layer_norm      <- the decomposed LayerNorm chain (ReduceMean :5237, Sub, Pow :5478, ReduceMean, Add, Sqrt, Div, Mul, Add).
//   osg_reduce_mean_last, osg_softmax_last <- ReduceMean (:5237-5393), XnnPack::softmax (:1958).
#include "osg_common.h"
#include "osg_gemm_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; i++) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; i++) t = fmaxf(t, red[i]);
    return t;
}

// ---- InstanceNormalization on [rows, L]: one block per row, three sweeps (the row stays in L2) ---------------------
template <typename T>
__global__ __launch_bounds__(1024) void instance_norm_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ bias, T* __restrict__ y, long L, int n_scale,
                                                             float eps) {
    __shared__ float red[16];
    const long row = blockIdx.x;
    const T* xr = x + row * L;
    T* yr = y + row * L;
    float s = 0.f;
    for (long i = threadIdx.x; i < L; i += blockDim.x) s += to_f32<T>(xr[i]);
    const float mean = block_sum(s, red) / (float)L;
    float q = 0.f;
    for (long i = threadIdx.x; i < L; i += blockDim.x) {
        float d = to_f32<T>(xr[i]) - mean;
        q += d * d;
    }
    const float var = block_sum(q, red) / (float)L;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float sc = scale ? scale[row % n_scale] : 1.f, bi = bias ? bias[row % n_scale] : 0.f;
    for (long i = threadIdx.x; i < L; i += blockDim.x) yr[i] = from_f32<T>(sc * ((to_f32<T>(xr[i]) - mean) * rstd) + bi);
}

// ---- fused GroupNorm, NHWC --------------------------------------------------------------------------------------
// pass 1: grid (S, N): block (s, n) sweeps its slab of the HW pixels over ALL channels with 16-byte coalesced loads; a
// thread keeps one fixed vector column (so each of its V elements belongs to a fixed group), accumulates (sum, sumsq)
// per element in registers, then folds them per group through LDS atomics and emits one (sum, sumsq) pair per group.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, long HW, int C, int G, int S) {
    constexpr int V = 8 / (sizeof(T) / 2);
    extern __shared__ float gsum[];  // [G][2] running group sums, then [R][cols*V][2] per-thread element sums of one column sweep
    const int s = blockIdx.x, n = blockIdx.y;
    const int cpg = C / G, cv = C / V;
    for (int i = threadIdx.x; i < G * 2; i += 256) gsum[i] = 0.f;
    const long p0 = HW * s / S, p1 = HW * (s + 1) / S;
    const T* base = x + (long)n * HW * C;
    const int cols = cv < 256 ? cv : 256;          // vector columns covered per sweep
    const int R = 256 / cols;                      // pixel rows covered per sweep
    const int tr = threadIdx.x / cols, tc = threadIdx.x - tr * cols;
    float* esum = gsum + 2 * G;                    // [R][cols*V][2]
    const int ew = cols * V;                       // channels covered per sweep
    for (int c0 = 0; c0 < cv; c0 += cols) {        // (one sweep unless C > 2048)
        const int c = c0 + tc;
        float sm[V], sq[V];
#pragma unroll
        for (int e = 0; e < V; e++) sm[e] = sq[e] = 0.f;
        if (tr < R && c < cv) {
            long p = p0 + tr;
            for (; p + 3L * R < p1; p += 4L * R) {   // 4 independent 16-byte loads in flight
                T xv[4][V];
#pragma unroll
                for (int u = 0; u < 4; u++) *reinterpret_cast<uint4*>(xv[u]) = *reinterpret_cast<const uint4*>(base + (p + (long)u * R) * C + (long)c * V);
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int e = 0; e < V; e++) {
                        float v = to_f32<T>(xv[u][e]);
                        sm[e] += v;
                        sq[e] += v * v;
                    }
            }
            for (; p < p1; p += R) {
                T xv[V];
                *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(base + p * C + (long)c * V);
#pragma unroll
                for (int e = 0; e < V; e++) {
                    float v = to_f32<T>(xv[e]);
                    sm[e] += v;
                    sq[e] += v * v;
                }
            }
        }
        // deterministic fold (no atomics: a float atomicAdd's order -- hence its rounding -- varies from launch to launch):
        // per-thread element sums -> LDS, then ONE thread per group adds its channels over the R rows in a fixed order
        if (tr < R) {
#pragma unroll
            for (int e = 0; e < V; e++) {
                esum[((tr * ew) + tc * V + e) * 2 + 0] = sm[e];
                esum[((tr * ew) + tc * V + e) * 2 + 1] = sq[e];
            }
        }
        __syncthreads();
        const int ch_lo = c0 * V, ch_hi = min(C, ch_lo + ew);
        for (int g = threadIdx.x; g < G; g += 256) {
            const int lo = max(g * cpg, ch_lo), hi = min((g + 1) * cpg, ch_hi);
            float as = 0.f, aq = 0.f;
            for (int ch = lo; ch < hi; ch++)
                for (int r = 0; r < R; r++) {
                    as += esum[((r * ew) + (ch - ch_lo)) * 2 + 0];
                    aq += esum[((r * ew) + (ch - ch_lo)) * 2 + 1];
                }
            if (lo < hi) {
                gsum[g * 2] += as;
                gsum[g * 2 + 1] += aq;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        part[(((long)n * G + g) * S + s) * 2 + 0] = gsum[g * 2];
        part[(((long)n * G + g) * S + s) * 2 + 1] = gsum[g * 2 + 1];
    }
}

// pass 2 (tiny): one block per image folds the partials and writes the
// per-channel affine table  y = x * ca[c] + cb[c]   (ca = rstd*gamma, cb = beta - mean*rstd*gamma)  to tab[n][2][C]
// fold of the S partials of every group of image n in f64 (fixed order => deterministic) -> stat[g] = (mean, rstd) in LDS
__device__ __forceinline__ void gn_fold_stats(const float* __restrict__ part, float* stat, int n, long HW, int cpg, int G, int S, float eps) {
    int tpg = 1;
    while (tpg * 2 * G <= 256 && tpg < 64) tpg *= 2;
    const double icnt = (double)(1.0f / ((float)HW * (float)cpg));
    for (int g0 = 0; g0 < G; g0 += 256 / tpg) {
        const int g = g0 + threadIdx.x / tpg, l = threadIdx.x % tpg;
        double sm = 0, q = 0;
        if (g < G)
            for (int k = l; k < S; k += tpg) {
                sm += part[(((long)n * G + g) * S + k) * 2 + 0];
                q += part[(((long)n * G + g) * S + k) * 2 + 1];
            }
        for (int o = tpg >> 1; o > 0; o >>= 1) {
            sm += __shfl_xor(sm, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (g < G && l == 0) {
            // E[x^2] - mean^2 in f64 (fma-rate operations; the f32 subtraction would cancel); the reciprocal count, the square root and the
            // final reciprocal in f32 -- an f64 divide / sqrt is ~100 instructions each, on the critical path of every block that folds
            const double mean = sm * icnt;
            const float var = fmaxf((float)(q * icnt - mean * mean), 0.f);
            stat[g * 2 + 0] = (float)mean;
            stat[g * 2 + 1] = 1.0f / sqrtf(var + eps);
        }
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const T* __restrict__ gamma, const T* __restrict__ beta,
                                                          float* __restrict__ tab, long HW, int C, int G, int S, float eps) {
    extern __shared__ float stat[];  // [G][2] mean, rstd
    const int n = blockIdx.x;
    const int cpg = C / G;
    gn_fold_stats(part, stat, n, HW, cpg, G, S, eps);
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        const float a = stat[g * 2 + 1] * to_f32<T>(gamma[c]);
        tab[((long)n * 2 + 0) * C + c] = a;
        tab[((long)n * 2 + 1) * C + c] = to_f32<T>(beta[c]) - stat[g * 2] * a;
    }
}

// pass 3: pure streaming  y = act(x * ca[c] + cb[c]).  A thread keeps ONE vector column (its 2*V table entries live in
// registers), sweeps pixel rows with 4 independent 16-byte loads in flight; grid (slabs, N).
// FOLD = 2 (round 3): the statistics come from the PRODUCER of x -- `part` is the StatSink table [N][G][2] of int64 fixed-point sums its epilogue filled
// (osg_gemm_common.h): mean / rstd of the block's image by the first G threads, then the same streaming pass
template <typename T, int FOLD>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ tab, T* __restrict__ y, long HW,
                                                       int C, int act, int slabs, const float* __restrict__ part, const T* __restrict__ gamma,
                                                       const T* __restrict__ beta, int G, int S, float eps) {
    osg_pin_all(x, tab, y, HW, C, act, slabs, part, gamma, beta, G, S, eps, (int)gridDim.y);
    constexpr int V = 8 / (sizeof(T) / 2);  // f16: 8, f32: 4  (16-byte accesses)
    extern __shared__ float stat[];         // FOLD: [G][2] mean, rstd -- pass 2 folded into every block's prologue (one launch fewer)
    const int n = blockIdx.y, sl = blockIdx.x;
    const int cpg = FOLD ? C / G : 1;
    if (FOLD == 1) gn_fold_stats(part, stat, n, HW, cpg, G, S, eps);
    if (FOLD == 2) {
        const long long* __restrict__ tb = reinterpret_cast<const long long*>(part) + (long)n * G * 2;
        const long copy_stride = (long)gridDim.y * G * 2;        // (gridDim.y = images)
        const double icnt = (double)(1.0f / ((float)HW * (float)cpg));
        for (int g = threadIdx.x; g < G; g += 256) {
            long long ts = 0, tq = 0;                            // the eight XCDs' copies (integers: any order gives the same bits)
#pragma unroll
            for (int k = 0; k < osg_mm::kStatCopies; k++) { ts += tb[k * copy_stride + g * 2]; tq += tb[k * copy_stride + g * 2 + 1]; }
            const double mean = (double)ts * (1.0 / (double)osg_mm::kStatSX) * icnt;
            const double q = (double)tq * (1.0 / (double)osg_mm::stat_q_scale((long)HW * cpg));
            const float var = fmaxf((float)(q * icnt - mean * mean), 0.f);
            stat[g * 2 + 0] = (float)mean;
            stat[g * 2 + 1] = 1.0f / sqrtf(var + eps);
        }
        __syncthreads();
    }
    const int cv = C / V;
    const int cols = cv < 256 ? cv : 256, R = 256 / cols;
    const int tr = threadIdx.x / cols, tc = threadIdx.x - tr * cols;
    if (tr >= R) return;
    const long p0 = HW * sl / slabs, p1 = HW * (sl + 1) / slabs;
    const T* xb = x + (long)n * HW * C;
    T* yb = y + (long)n * HW * C;
    const float* ta = tab + (long)n * 2 * C;
    for (int c = tc; c < cv; c += cols) {
        float ca[V], cb[V];
#pragma unroll
        for (int e = 0; e < V; e++) {
            if (FOLD) {     // the very expressions of gn_finalize_kernel: identical bits
                const int ch = c * V + e, g = ch / cpg;
                const float a = stat[g * 2 + 1] * to_f32<T>(gamma[ch]);
                ca[e] = a;
                cb[e] = to_f32<T>(beta[ch]) - stat[g * 2] * a;
            } else {
                ca[e] = ta[c * V + e];
                cb[e] = ta[C + c * V + e];
            }
        }
        long p = p0 + tr;
        for (; p + 3L * R < p1; p += 4L * R) {
            T xv[4][V], ov[V];
#pragma unroll
            for (int u = 0; u < 4; u++) *reinterpret_cast<uint4*>(xv[u]) = *reinterpret_cast<const uint4*>(xb + (p + (long)u * R) * C + (long)c * V);
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int e = 0; e < V; e++) ov[e] = from_f32<T>(osg_apply_act(to_f32<T>(xv[u][e]) * ca[e] + cb[e], act));
                *reinterpret_cast<uint4*>(yb + (p + (long)u * R) * C + (long)c * V) = *reinterpret_cast<uint4*>(ov);
            }
        }
        for (; p < p1; p += R) {
            T xv[V], ov[V];
            *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(xb + p * C + (long)c * V);
#pragma unroll
            for (int e = 0; e < V; e++) ov[e] = from_f32<T>(osg_apply_act(to_f32<T>(xv[e]) * ca[e] + cb[e], act));
            *reinterpret_cast<uint4*>(yb + p * C + (long)c * V) = *reinterpret_cast<uint4*>(ov);
        }
    }
}

// ---- GroupNorm, one launch: block (gblk, n) owns `gb` whole groups (= CW contiguous channels, CW % 8 == 0) of one image and keeps its
// [HW x CW] slab in registers between the statistics and the apply: x is read once, y written once, no workspace, no second launch.
// The UNet's 8x8 / 16x16 / 32x32 levels (45 of SD1.5's 61 GroupNorms) are launch-latency bound, not bandwidth bound, and run this way;
// larger slabs use the three-pass path below.  Thread t keeps ONE vector column (8 channels => at most two groups when cpg >= 8).
// CL (cluster): blockIdx.z = s splits the pixel rows of the slab over S co-resident blocks (the 64x64 level: 16 x 2 x 8 blocks instead of a
// statistics launch and an apply launch that reads x again).  Each block publishes its per-group (sum, sumsq) write-through (sc1), takes a
// relaxed agent-scope ticket, waits until all S partials of its slab are there, folds them in slab order (=> the same bits in every block) and
// applies from registers.  The grid never exceeds the CU count, so on an idle GPU every block is resident while its peers wait -- but the wait is
// BOUNDED (200 us by default, OSG_GN_CLUSTER_WAIT in 10 ns ticks): a block whose peers do not show up (another context / stream holds their CUs)
// recomputes their partials from x itself, with the same reduction tree => the same bits, and no deadlock whatever else runs on the device.
// The last block to LEAVE a cluster zeroes its two counters for the next launch.
template <int NV, bool CL>
__global__ __launch_bounds__(1024) void gn_slab_kernel(const f16* __restrict__ x, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                       f16* __restrict__ y, int HW, int C, int cpg, int gb, float eps, int act, int S,
                                                       double* __restrict__ part, int* __restrict__ cnt, int wait_ticks, long long* __restrict__ kdbg, int NT,
                                                       int gdx, int gdy) {
    // NT / gdx / gdy = blockDim.x / gridDim.x / gridDim.y as ARGUMENTS (round 6): the implicit ones cost the prologue a vector load with a full drain behind it
    osg_pin_all(x, gamma, beta, y, HW, C, cpg, gb, eps, act, S, part, cnt, wait_ticks, kdbg, NT, gdx, gdy);
    auto stamp = [&](int slot) { if (__builtin_expect(kdbg != nullptr, 0) && threadIdx.x == 0) kdbg[((long)(blockIdx.z * gdy + blockIdx.y) * gdx + blockIdx.x) * 8 + slot] = wall_clock64(); };
    stamp(0);
    __shared__ float red[8][16][2];   // [local group][wave][sum, sumsq]
    __shared__ float stat[8][2];      // [local group][mean, rstd]
    const int HWs = CL ? HW / S : HW;      // rows of this block
    const int nw = NT >> 6;
    const int CW = gb * cpg, VR = CW >> 3, RT = NT / VR;
    const int t = threadIdx.x, tr = t / VR, tc = t - tr * VR;
    const bool active = tr < RT;
    const int ch = blockIdx.x * CW + tc * 8;                 // first of this thread's 8 channels
    const long img = (long)blockIdx.y * HW * C + blockIdx.x * CW + (CL ? (long)blockIdx.z * HWs * C : 0);   // uniform base; per-thread offsets stay 32-bit (scalar base + voffset loads)
    const f16* xb = x + img;
    f16* yb = y + img;
    const unsigned off0 = (unsigned)tr * C + tc * 8, step = (unsigned)RT * C;
    f16x8 v[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int row = tr + j * RT;
        if (active && row < HWs) v[j] = *reinterpret_cast<const f16x8*>(xb + (off0 + j * step));
        else v[j] = (f16x8)(f16)0;
    }
    stamp(1);
    float sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; e++) sm[e] = sq[e] = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float f = (float)v[j][e];
            sm[e] += f;
            sq[e] = fmaf(f, f, sq[e]);
        }
    stamp(2);
    // keep the slab PACKED (f16) across the reduction: without this the compiler holds the converted f32 copies live (2x the registers)
#pragma unroll
    for (int j = 0; j < NV; j++) asm volatile("" : "+v"(v[j]));
    // elements [0, esplit) belong to local group g0, the rest to g0 + 1
    const int g0 = (tc * 8) / cpg;
    const int esplit = (g0 + 1) * cpg - tc * 8;
    float as = 0.f, aq = 0.f, bs = 0.f, bq = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (e < esplit) { as += sm[e]; aq += sq[e]; }
        else            { bs += sm[e]; bq += sq[e]; }
    }
    // deterministic block reduction per local group: xor-shuffle tree inside a wave, then a fixed-order sum over the waves
    const int lane = t & 63, wave = t >> 6;
    for (int gl = 0; gl < gb; gl++) {
        float s = active ? (g0 == gl ? as : (g0 + 1 == gl ? bs : 0.f)) : 0.f;
        float q = active ? (g0 == gl ? aq : (g0 + 1 == gl ? bq : 0.f)) : 0.f;
        s = wave_sum_valu(s);
        q = wave_sum_valu(q);
        if (lane == 0) { red[gl][wave][0] = s; red[gl][wave][1] = q; }
    }
    __syncthreads();
    stamp(3);
    if constexpr (CL) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        __shared__ int alone;             // 1: the wait for the peers ran out -- this block computes every partial of its slab itself
        const long slab = (long)blockIdx.y * gdx + blockIdx.x;
        double* mine = part + ((slab * gb) * S) * 2;
        int* c2 = cnt + slab * 2;
        d2 own = {0.0, 0.0};              // (threads t < gb) this block's partial of local group t
        if (t < gb) {
            for (int w = 0; w < nw; w++) { own[0] += red[t][w][0]; own[1] += red[t][w][1]; }
            double* dst = mine + ((long)t * S + blockIdx.z) * 2;
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(own) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (t == 0) {
            // BOUNDED wait (advisor, round 2): co-residency of the S blocks of a slab is only guaranteed on an otherwise idle GPU -- a second context or a
            // second stream can hold the CUs some peers need.  After `wait_ticks` (10 ns each) without the full count the block stops waiting and
            // recomputes its peers' partials from x itself (below): slower, same bits, and nobody ever spins without a bound.
            __hip_atomic_fetch_add(c2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            bool all;
            do {
                all = __hip_atomic_load(c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= S;
                if (!all) __builtin_amdgcn_s_sleep(1);
            } while (!all && wall_clock64() - t0 < (unsigned long long)wait_ticks);
            alone = all ? 0 : 1;
        }
        __syncthreads();
        stamp(4);
        const bool solo = alone != 0;
        double s = 0, q = 0;
        if (__builtin_expect(!solo, 1)) {
            if (t < gb) {
                for (int s0 = 0; s0 < S; s0 += 4) {
                    d2 pv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double* a = mine + ((long)t * S + min(s0 + u, S - 1)) * 2;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[u]) : "v"(a) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        asm volatile("" : "+v"(pv[u]));
                        if (s0 + u < S) { s += pv[u][0]; q += pv[u][1]; }
                    }
                }
            }
        } else {
            // the partial of part z, computed exactly as the block that owns it does (same per-thread row order, same reduction tree): identical bits
            for (int z = 0; z < S; z++) {
                d2 pz = own;
                if (z != (int)blockIdx.z) {   // (uniform over the block)
                    const f16* xz = x + (long)blockIdx.y * HW * C + blockIdx.x * CW + (long)z * HWs * C;
                    float zm[8], zq[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) zm[e] = zq[e] = 0.f;
#pragma unroll 1   // (the rare path must not cost the common one registers: one row vector in flight at a time)
                    for (int j = 0; j < NV; j++) {
                        const int row = tr + j * RT;
                        f16x8 w8 = (f16x8)(f16)0;
                        if (active && row < HWs) w8 = *reinterpret_cast<const f16x8*>(xz + (off0 + j * step));
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float f = (float)w8[e];
                            zm[e] += f;
                            zq[e] = fmaf(f, f, zq[e]);
                        }
                    }
                    float zas = 0.f, zaq = 0.f, zbs = 0.f, zbq = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        if (e < esplit) { zas += zm[e]; zaq += zq[e]; }
                        else            { zbs += zm[e]; zbq += zq[e]; }
                    }
                    __syncthreads();          // red[] of the previous part has been consumed
                    for (int gl = 0; gl < gb; gl++) {
                        float ps = active ? (g0 == gl ? zas : (g0 + 1 == gl ? zbs : 0.f)) : 0.f;
                        float pq2 = active ? (g0 == gl ? zaq : (g0 + 1 == gl ? zbq : 0.f)) : 0.f;
                        ps = wave_sum_valu(ps);
                        pq2 = wave_sum_valu(pq2);
                        if (lane == 0) { red[gl][wave][0] = ps; red[gl][wave][1] = pq2; }
                    }
                    __syncthreads();
                    pz = d2{0.0, 0.0};
                    if (t < gb)
                        for (int w = 0; w < nw; w++) { pz[0] += red[t][w][0]; pz[1] += red[t][w][1]; }
                }
                if (t < gb) { s += pz[0]; q += pz[1]; }
            }
        }
        if (t < gb) {
            const double icnt = (double)(1.0f / ((float)HW * (float)cpg));
            const double mean = s * icnt;
            const float var = fmaxf((float)(q * icnt - mean * mean), 0.f);
            stat[t][0] = (float)mean;
            stat[t][1] = 1.0f / sqrtf(var + eps);
        }
        __syncthreads();
        if (t == 0) {   // (every reader of this cluster's partials is past its loads when the last one leaves; a block that went solo reads none)
            if (__hip_atomic_fetch_add(c2 + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S - 1) {
                __hip_atomic_store(c2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c2 + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
    if (t < gb) {
        double s = 0, q = 0;
        for (int w = 0; w < nw; w++) { s += red[t][w][0]; q += red[t][w][1]; }
        const double icnt = (double)(1.0f / ((float)HW * (float)cpg));   // (f64 only where the cancellation is; see gn_fold_stats)
        const double mean = s * icnt;
        const float var = fmaxf((float)(q * icnt - mean * mean), 0.f);
        stat[t][0] = (float)mean;
        stat[t][1] = 1.0f / sqrtf(var + eps);
    }
    __syncthreads();
    }
    stamp(5);
    if (!active) return;
    float ca[8], cb[8];
    {
        const f16x8 ga = *reinterpret_cast<const f16x8*>(gamma + ch);
        const f16x8 be = *reinterpret_cast<const f16x8*>(beta + ch);
        const float m0 = stat[g0][0], r0 = stat[g0][1];
        const int g1 = g0 + 1 < gb ? g0 + 1 : g0;
        const float m1 = stat[g1][0], r1 = stat[g1][1];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float a = (e < esplit ? r0 : r1) * (float)ga[e];
            ca[e] = a;
            cb[e] = (float)be[e] - (e < esplit ? m0 : m1) * a;
        }
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int row = tr + j * RT;
        if (row < HWs) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (f16)osg_apply_act(fmaf((float)v[j][e], ca[e], cb[e]), act);
            *reinterpret_cast<f16x8*>(yb + (off0 + j * step)) = o;
        }
    }
    stamp(6);
    if (__builtin_expect(kdbg != nullptr, 0)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(7); }
}

// launch plan of the slab kernel: groups per block (smallest gb with gb*cpg % 8 == 0), threads, vectors per thread; false => three-pass path
struct GnSlabPlan { int gb, nt, nv; };
bool gn_slab_plan(long HW, int C, int G, GnSlabPlan* pl) {
    const bool off = getenv("OSG_GN_SLAB_OFF") != nullptr;  // (tests: pin the three-pass statistics)
    if (off || G <= 0 || C % G || C % 8) return false;
    const int cpg = C / G;
    if (cpg < 8) return false;
    int gb = 1;
    while (gb <= 8 && ((gb * cpg) % 8 || G % gb)) gb++;
    if (gb > 8) return false;
    const int VR = gb * cpg / 8;
    if (VR > 128 || HW > (1 << 20)) return false;
    for (int nt = 256; nt <= 1024; nt *= 2) {
        const int RT = nt / VR;
        if (RT < 1) continue;
        const long need = (HW + RT - 1) / RT;
        const long cap = nt < 1024 ? 4 : 8;   // (16 vectors per thread spills and is no faster than the three-pass path)
        if (need <= cap) {
            int nv = 1;
            while (nv < need) nv *= 2;
            *pl = {gb, nt, nv};
            return true;
        }
    }
    return false;
}

// cluster plan (slabs too large for one block's registers): the largest S in {16, 8, 4, 2} whose grid still fits the CUs, then the smallest block
struct GnClusterPlan { int gb, nt, nv, S; };
bool gn_cluster_plan(long HW, int C, int G, int N, int num_cu, GnClusterPlan* pl) {
    const bool off = getenv("OSG_GN_CLUSTER_OFF") != nullptr;   // (tests: pin the three-pass statistics)
    if (off || G <= 0 || C % G || C % 8) return false;
    const int cpg = C / G;
    if (cpg < 8) return false;
    int gb = 1;
    while (gb <= 8 && ((gb * cpg) % 8 || G % gb)) gb++;
    if (gb > 8) return false;
    const int VR = gb * cpg / 8;
    if (VR > 128 || HW > (1 << 20)) return false;
    const long slabs = (long)(G / gb) * N;
    if (slabs * 2 > osg_ctx::kTickets / 2) return false;
    for (int S = 16; S >= 2; S >>= 1) {
        if (slabs * S > num_cu || HW % S) continue;
        const long HWs = HW / S;
        for (int nt = 256; nt <= 1024; nt *= 2) {
            const int RT = nt / VR;
            if (RT < 1) continue;
            const long need = (HWs + RT - 1) / RT;
            if (need <= (nt < 1024 ? 4 : 8)) {
                int nv = 1;
                while (nv < need) nv *= 2;
                *pl = {gb, nt, nv, S};
                return true;
            }
        }
    }
    return false;
}

// ---- LayerNorm over the last axis: one wave per row, row cached in registers (C <= 64*8*MAXV) --------------------
template <typename T, int NV>
__global__ __launch_bounds__(256) void layer_norm_kernel(const T* __restrict__ x, const T* __restrict__ gamma, const T* __restrict__ beta,
                                                         T* __restrict__ y, long rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * C;
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        int c = (j * 64 + lane) * 8;
        if (c < C) {
            f16x8 t = *reinterpret_cast<const f16x8*>(xr + c);
#pragma unroll
            for (int e = 0; e < 8; e++) { v[j][e] = (float)t[e]; s += v[j][e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[j][e] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        int c = (j * 64 + lane) * 8;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 8; e++) { float d = v[j][e] - mean; q += d * d; }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; j++) {
        int c = (j * 64 + lane) * 8;
        if (c < C) {
            f16x8 g = *reinterpret_cast<const f16x8*>(gamma + c);
            f16x8 b = *reinterpret_cast<const f16x8*>(beta + c);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (f16)((v[j][e] - mean) * rstd * (float)g[e] + (float)b[e]);
            *reinterpret_cast<f16x8*>(y + row * C + c) = o;
        }
    }
}

// RMSNorm, one block per row (any C): sum of squares in fp32, y = f16(w * (x * (1 / sqrt(mean + eps))))
__global__ __launch_bounds__(256) void rms_norm_kernel(const f16* __restrict__ x, const f16* __restrict__ w, f16* __restrict__ y, int C, float eps) {
    __shared__ float red[8];
    const f16* xr = x + (long)blockIdx.x * C;
    f16* yr = y + (long)blockIdx.x * C;
    float ss = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = (float)xr[c];
        ss += v * v;
    }
    ss = block_sum(ss, red);
    const float r = 1.0f / sqrtf(ss / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) yr[c] = (f16)((float)w[c] * ((float)xr[c] * r));
}

// generic (any C / dtype): one block per row
template <typename T>
__global__ __launch_bounds__(256) void layer_norm_generic_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                                 const T* __restrict__ beta, T* __restrict__ y, int C, float eps) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const T* xr = x + row * C;
    float s = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) s += to_f32<T>(xr[i]);
    const float mean = block_sum(s, red) / (float)C;
    float q = 0.f;
    for (int i = threadIdx.x; i < C; i += 256) { float d = to_f32<T>(xr[i]) - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / (float)C + eps);
    for (int i = threadIdx.x; i < C; i += 256)
        y[row * C + i] = from_f32<T>((to_f32<T>(xr[i]) - mean) * rstd * to_f32<T>(gamma[i]) + to_f32<T>(beta[i]));
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_mean_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, long C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * C;
    float s = 0.f;
    for (long i = lane; i < C; i += 64) s += to_f32<T>(xr[i]);
    s = wave_sum(s);
    if (lane == 0) y[row] = from_f32<T>(s / (float)C);
}

// Softmax over the last axis; restates XNNPACK's three passes (rmax / raddstoreexpminusmax / vmulc):
// e = exp(x-max) stored in T, sum from the un-rounded e, out = T(e) * T(1/sum).
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(const T* __restrict__ x, T* __restrict__ y, long C) {
    __shared__ float red[8];
    const long row = blockIdx.x;
    const T* xr = x + row * C;
    T* yr = y + row * C;
    float mx = -INFINITY;
    for (long i = threadIdx.x; i < C; i += 256) mx = fmaxf(mx, to_f32<T>(xr[i]));
    mx = block_max(mx, red);
    float s = 0.f;
    for (long i = threadIdx.x; i < C; i += 256) s += expf(to_f32<T>(xr[i]) - mx);
    s = block_sum(s, red);
    const float rinv = to_f32<T>(from_f32<T>(1.0f / s));
    for (long i = threadIdx.x; i < C; i += 256) {
        float e = to_f32<T>(from_f32<T>(expf(to_f32<T>(xr[i]) - mx)));
        yr[i] = from_f32<T>(e * rinv);
    }
}

// GroupNorm from the producer's statistics (StatSink tables), the latency-ordered form: a launch of the UNet pass lasts as long as one workgroup, so the
// workgroup's own chain is what counts.  Every thread first REQUESTS its rows of x (U x 16 bytes) and its channels' gamma / beta -- none of which depends on
// the statistics -- then the first G threads turn the eight table copies into mean / rstd while those loads are in flight, one barrier, apply, store.
// (gn_apply_kernel<f16, 2> has the table reads and the barrier in front of the first x load.  Measured: no better -- see osg_group_norm_stats_nhwc.)
// cols = C / 8 vector columns (<= 256), R = 256 / cols rows per sweep, U sweeps per workgroup.
template <int U>
__global__ __launch_bounds__(256) void gn_apply_stats_kernel(const f16* __restrict__ x, const long long* __restrict__ table, f16* __restrict__ y, long HW, int C, int act,
                                                             const f16* __restrict__ gamma, const f16* __restrict__ beta, int G, float eps) {
    osg_pin_all(x, table, y, HW, C, act, gamma, beta, G, eps, (int)gridDim.y);
    __shared__ float stat[512];            // [G][2] mean, rstd (G <= 256)
    const int n = blockIdx.y, cv = C >> 3, R = 256 / cv;
    const int tr = threadIdx.x / cv, tc = threadIdx.x - tr * cv;
    const bool live = tr < R;
    const long p0 = (long)blockIdx.x * R * U + tr;
    const f16* xb = x + (long)n * HW * C + (long)tc * 8;
    f16* yb = y + (long)n * HW * C + (long)tc * 8;
    uint4 xv[U], gv = {0, 0, 0, 0}, bv = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; u++) {
        const long p = p0 + (long)u * R;
        xv[u] = (live && p < HW) ? *reinterpret_cast<const uint4*>(xb + p * C) : uint4{0, 0, 0, 0};
    }
    if (live) {
        gv = *reinterpret_cast<const uint4*>(gamma + tc * 8);
        bv = *reinterpret_cast<const uint4*>(beta + tc * 8);
    }
    const int cpg = C / G;
    {
        const long long* __restrict__ tb = table + (long)n * G * 2;
        const long copy_stride = (long)gridDim.y * G * 2;
        const double icnt = (double)(1.0f / ((float)HW * (float)cpg));
        for (int g = threadIdx.x; g < G; g += 256) {
            long long ts = 0, tq = 0;                            // the eight XCDs' copies (integers: any order gives the same bits)
#pragma unroll
            for (int k = 0; k < osg_mm::kStatCopies; k++) { ts += tb[k * copy_stride + g * 2]; tq += tb[k * copy_stride + g * 2 + 1]; }
            const double mean = (double)ts * (1.0 / (double)osg_mm::kStatSX) * icnt;
            const double q = (double)tq * (1.0 / (double)osg_mm::stat_q_scale((long)HW * cpg));
            const float var = fmaxf((float)(q * icnt - mean * mean), 0.f);
            stat[g * 2 + 0] = (float)mean;
            stat[g * 2 + 1] = 1.0f / sqrtf(var + eps);
        }
    }
    __syncthreads();
    if (!live) return;
    float ca[8], cb[8];
    const f16* g8 = reinterpret_cast<const f16*>(&gv);
    const f16* b8 = reinterpret_cast<const f16*>(&bv);
#pragma unroll
    for (int e = 0; e < 8; e++) {           // the expressions of gn_apply_kernel
        const int g = (tc * 8 + e) / cpg;
        const float a = stat[g * 2 + 1] * (float)g8[e];
        ca[e] = a;
        cb[e] = (float)b8[e] - stat[g * 2] * a;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const long p = p0 + (long)u * R;
        if (p >= HW) break;
        const f16* xe = reinterpret_cast<const f16*>(&xv[u]);
        uint4 ov;
        f16* oe = reinterpret_cast<f16*>(&ov);
#pragma unroll
        for (int e = 0; e < 8; e++) oe[e] = (f16)osg_apply_act((float)xe[e] * ca[e] + cb[e], act);
        *reinterpret_cast<uint4*>(yb + p * C) = ov;
    }
}

// StatSink statistics from a stored output (the launches whose epilogue does not serve sinks: split-K, ragged shapes, the small-Cin convolution): a
// workgroup owns 128 rows x 64 channels -- thread = (channel, one of four 32-row parts) -- and adds per group what the fused epilogue would have added
__global__ __launch_bounds__(256) void colstats_kernel(const f16* __restrict__ C, long ldc, int M, int N, int hw, osg_mm::StatSink s0, osg_mm::StatSink s1, int imgs, int per_xcd) {
    __shared__ float st[4][64][2];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 64, n = n0 + c;
    float s = 0.f, q = 0.f;
    if (n < N)
        for (int r = part * 32; r < part * 32 + 32; r++) {
            const int m = m0 + r;
            if (m >= M) break;
            const float f = (float)C[(long)m * ldc + n];
            s += f;
            q = fmaf(f, f, q);
        }
    st[part][c][0] = s;
    st[part][c][1] = q;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x, wn = min(64, N - n0), n_img = m0 / hw;
    const osg_mm::StatSink sk[2] = {s0, s1};
    for (int k = 0; k < 2; k++) {
        if (!sk[k].table) continue;
        const int c_lo = n0 + sk[k].ch_off, c_hi = c_lo + wn;
        const int g = c_lo / sk[k].cpg + lane;
        if (g * sk[k].cpg < c_hi) {
            const int a = max(g * sk[k].cpg, c_lo) - c_lo, b = min((g + 1) * sk[k].cpg, c_hi) - c_lo;
            float S = 0.f, Q = 0.f;
            for (int cc = a; cc < b; cc++)
                for (int pp = 0; pp < 4; pp++) { S += st[pp][cc][0]; Q += st[pp][cc][1]; }
            osg_mm::stat_add(per_xcd, reinterpret_cast<unsigned long long*>(sk[k].table), (long)imgs * sk[k].groups * 2, ((long)n_img * sk[k].groups + g) * 2, S, Q, osg_mm::stat_q_scale((long)hw * sk[k].cpg));
        }
    }
}

}  // namespace

int osg_mm::launch_colstats(osg_ctx* ctx, const f16* C, long ldc, int M, int N, int rows_per_image, const StatSink* sinks) {
    if (rows_per_image <= 0 || rows_per_image % 128 || M % rows_per_image) OSG_FAIL(ctx, "statistics sinks: the image size must be a multiple of 128 rows");
    hipLaunchKernelGGL(colstats_kernel, dim3((unsigned)((M + 127) / 128), (unsigned)((N + 63) / 64)), dim3(256), 0, ctx->compute, C, ldc, M, N, rows_per_image, sinks[0], sinks[1], M / rows_per_image, ctx->xcd_ids8 ? 1 : 0);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

extern "C" {

int osg_instance_norm(osg_ctx* ctx, osg_dtype dtype, const void* x, const float* scale, const float* bias, void* y, int rows, long L,
                      int n_scale, float eps) {
    if (rows <= 0 || L <= 0) return 0;
    if (n_scale <= 0) n_scale = 1;
    int threads = L >= 8192 ? 1024 : 256;
    if (dtype == OSG_F16)
        hipLaunchKernelGGL(instance_norm_kernel<f16>, dim3(rows), dim3(threads), 0, ctx->compute, (const f16*)x, scale, bias, (f16*)y, L,
                           n_scale, eps);
    else if (dtype == OSG_F32)
        hipLaunchKernelGGL(instance_norm_kernel<float>, dim3(rows), dim3(threads), 0, ctx->compute, (const float*)x, scale, bias,
                           (float*)y, L, n_scale, eps);
    else
        OSG_FAIL(ctx, "osg_instance_norm: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"


extern "C" {

// GroupNorm whose statistics the producer of x has already added up (osg_set_stat_sinks -> the convolution's epilogue): table [N][G][2] int64, see
// osg_gemm_common.h StatSink.  One streaming launch: every workgroup turns its image's sums into mean / rstd and normalises its rows.
int osg_group_norm_stats_nhwc(osg_ctx* ctx, const void* x, const void* gamma, const void* beta, void* y, int N, long HW, int C, int G, float eps, osg_act act,
                              const void* stat_table) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    if (G <= 0 || C % G || C % 8 || !stat_table) OSG_FAIL(ctx, "osg_group_norm_stats_nhwc: invalid argument");
    const int cv = C / 8;
    // (measured: the request-first order is NOT faster inside the pass -- 5.81 vs 5.59 ms per step with the plain order on comparable boxes,
    // profiles/r03_gn_stats_ab.txt; it stays behind OSG_GN_STATS_APPLY_V2=1)
    static const bool v2 = getenv("OSG_GN_STATS_APPLY_V2") != nullptr;
    if (cv <= 256 && G <= 256 && v2) {
        const int R = 256 / cv;
        constexpr int U = 4;
        const unsigned blocks = (unsigned)((HW + (long)R * U - 1) / ((long)R * U));
        hipLaunchKernelGGL((gn_apply_stats_kernel<U>), dim3(blocks, N), dim3(256), 0, ctx->compute, (const f16*)x, (const long long*)stat_table, (f16*)y, HW, C, (int)act,
                           (const f16*)gamma, (const f16*)beta, G, eps);
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    const int cols = cv < 256 ? cv : 256, R = 256 / cols;
    long slabs = HW / ((long)R * 4);
    if (slabs < 1) slabs = 1;
    if (slabs > 4096) slabs = 4096;
    hipLaunchKernelGGL((gn_apply_kernel<f16, 2>), dim3((unsigned)slabs, N), dim3(256), G * 2 * sizeof(float), ctx->compute, (const f16*)x, (const float*)nullptr, (f16*)y,
                       HW, C, (int)act, (int)slabs, (const float*)stat_table, (const f16*)gamma, (const f16*)beta, G, 1, eps);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_group_norm_nhwc(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* gamma, const void* beta, void* y, int N, long HW,
                        int C, int G, float eps, osg_act act) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    if (G <= 0 || C % G) OSG_FAIL(ctx, "osg_group_norm_nhwc: C must be a multiple of groups");
    const int V = dtype == OSG_F16 ? 8 : 4;
    if (dtype != OSG_F16 && dtype != OSG_F32) OSG_FAIL(ctx, "osg_group_norm_nhwc: unsupported dtype");
    if (C % V) OSG_FAIL(ctx, "osg_group_norm_nhwc: C must be a multiple of the 16-byte vector width");
    GnSlabPlan sp;
    // OSG_GN_CLUSTER_MIN_NV=n: slabs that need >= n row vectors per thread in the one-block kernel (few, fat blocks: 32 x 1024 threads at the 32x32
    // level) run as clusters instead when a cluster plan exists (A/B knob, round 3)
    const int cl_min_nv = getenv("OSG_GN_CLUSTER_MIN_NV") ? atoi(getenv("OSG_GN_CLUSTER_MIN_NV")) : 1 << 20;
    GnClusterPlan cp0;
    if (dtype == OSG_F16 && gn_slab_plan(HW, C, G, &sp) && !(sp.nv >= cl_min_nv && ctx->tickets && gn_cluster_plan(HW, C, G, N, ctx->num_cu, &cp0))) {
        const dim3 grid(G / sp.gb, N), block(sp.nt);
#define OSG_GN_SLAB(NV_)                                                                                                             \
    hipLaunchKernelGGL((gn_slab_kernel<NV_, false>), grid, block, 0, ctx->compute, (const f16*)x, (const f16*)gamma, (const f16*)beta, (f16*)y, \
                       (int)HW, C, C / G, sp.gb, eps, (int)act, 1, (double*)nullptr, (int*)nullptr, 0, osg_mm::kdbg_buffer(ctx, (long)grid.x * grid.y), \
                       (int)block.x, (int)grid.x, (int)grid.y)
        switch (sp.nv) {
            case 1: OSG_GN_SLAB(1); break;
            case 2: OSG_GN_SLAB(2); break;
            case 4: OSG_GN_SLAB(4); break;
            default: OSG_GN_SLAB(8); break;
        }
#undef OSG_GN_SLAB
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    GnClusterPlan cp;
    if (dtype == OSG_F16 && ctx->tickets && gn_cluster_plan(HW, C, G, N, ctx->num_cu, &cp)) {
        if (osg_ensure_workspace(ctx, (size_t)N * G * cp.S * 2 * sizeof(double))) return 1;
        const dim3 grid(G / cp.gb, N, cp.S), block(cp.nt);
        int* cnt = ctx->tickets + osg_ctx::kTickets / 2;   // (the lower half belongs to the split-K tickets)
        const int gn_wait = getenv("OSG_GN_CLUSTER_WAIT") ? atoi(getenv("OSG_GN_CLUSTER_WAIT")) : 20000;   // 10 ns ticks; 0 = nobody waits (tests: every block goes solo)
#define OSG_GN_CL(NV_)                                                                                                               \
    hipLaunchKernelGGL((gn_slab_kernel<NV_, true>), grid, block, 0, ctx->compute, (const f16*)x, (const f16*)gamma, (const f16*)beta, (f16*)y, \
                       (int)HW, C, C / G, cp.gb, eps, (int)act, cp.S, (double*)ctx->ws, cnt, gn_wait, osg_mm::kdbg_buffer(ctx, (long)grid.x * grid.y * grid.z), \
                       (int)block.x, (int)grid.x, (int)grid.y)
        switch (cp.nv) {
            case 1: OSG_GN_CL(1); break;
            case 2: OSG_GN_CL(2); break;
            case 4: OSG_GN_CL(4); break;
            default: OSG_GN_CL(8); break;
        }
#undef OSG_GN_CL
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    // slabs of >= ~8 pixel rows per sweep-row, at most 64 per image (the apply pass folds S partials per group)
    const int cols = C / V < 256 ? C / V : 256;
    const int Rr = 256 / cols;
    int S = (int)(HW / ((long)Rr * 8));
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    const size_t part_bytes = ((size_t)N * G * S * 2 * sizeof(float) + 255) & ~(size_t)255;
    const size_t need = part_bytes + (size_t)N * 2 * C * sizeof(float);
    if (osg_ensure_workspace(ctx, need)) return 1;
    float* part = (float*)ctx->ws;
    float* tab = (float*)((char*)ctx->ws + part_bytes);
    // apply: ~8 pixel rows per thread-row
    long slabs_l = HW / ((long)Rr * 8);
    if (slabs_l < 1) slabs_l = 1;
    if (slabs_l > 4096) slabs_l = 4096;
    const int slabs = (int)slabs_l;
    if (dtype == OSG_F16) {
        hipLaunchKernelGGL(gn_stats_kernel<f16>, dim3(S, N), dim3(256), (G * 2 + 2 * 256 * 8) * sizeof(float), ctx->compute, (const f16*)x, part, HW, C, G, S);
        OSG_LAUNCH_CHECK(ctx);
        if ((long)slabs * N <= 1024) {
            // few apply blocks (the UNet's 64x64 level): each folds the partials itself -- 16 KB of L2 reads buy one launch less
            hipLaunchKernelGGL((gn_apply_kernel<f16, true>), dim3(slabs, N), dim3(256), G * 2 * sizeof(float), ctx->compute, (const f16*)x, tab, (f16*)y,
                               HW, C, (int)act, slabs, part, (const f16*)gamma, (const f16*)beta, G, S, eps);
        } else {
            hipLaunchKernelGGL(gn_finalize_kernel<f16>, dim3(N), dim3(256), G * 2 * sizeof(float), ctx->compute, part, (const f16*)gamma,
                               (const f16*)beta, tab, HW, C, G, S, eps);
            OSG_LAUNCH_CHECK(ctx);
            hipLaunchKernelGGL((gn_apply_kernel<f16, false>), dim3(slabs, N), dim3(256), 0, ctx->compute, (const f16*)x, tab, (f16*)y, HW, C, (int)act,
                               slabs, nullptr, nullptr, nullptr, G, S, eps);
        }
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(S, N), dim3(256), (G * 2 + 2 * 256 * 4) * sizeof(float), ctx->compute, (const float*)x, part, HW, C, G, S);
        OSG_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(N), dim3(256), G * 2 * sizeof(float), ctx->compute, part, (const float*)gamma,
                           (const float*)beta, tab, HW, C, G, S, eps);
        OSG_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL((gn_apply_kernel<float, false>), dim3(slabs, N), dim3(256), 0, ctx->compute, (const float*)x, tab, (float*)y, HW, C, (int)act,
                           slabs, nullptr, nullptr, nullptr, G, S, eps);
    }
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_layer_norm(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* gamma, const void* beta, void* y, long rows, int C,
                   float eps) {
    if (rows <= 0 || C <= 0) return 0;
    if (dtype == OSG_F16 && C % 8 == 0 && C <= 64 * 8 * 4) {
        dim3 grid((unsigned)((rows + 3) / 4)), block(256);
        int nv = (C + 511) / 512;
#define OSG_LN(NV) hipLaunchKernelGGL((layer_norm_kernel<f16, NV>), grid, block, 0, ctx->compute, (const f16*)x, (const f16*)gamma, \
                                      (const f16*)beta, (f16*)y, rows, C, eps)
        if (nv == 1) OSG_LN(1);
        else if (nv == 2) OSG_LN(2);
        else if (nv == 3) OSG_LN(3);
        else OSG_LN(4);
#undef OSG_LN
    } else if (dtype == OSG_F16) {
        hipLaunchKernelGGL(layer_norm_generic_kernel<f16>, dim3((unsigned)rows), dim3(256), 0, ctx->compute, (const f16*)x,
                           (const f16*)gamma, (const f16*)beta, (f16*)y, C, eps);
    } else if (dtype == OSG_F32) {
        hipLaunchKernelGGL(layer_norm_generic_kernel<float>, dim3((unsigned)rows), dim3(256), 0, ctx->compute, (const float*)x,
                           (const float*)gamma, (const float*)beta, (float*)y, C, eps);
    } else
        OSG_FAIL(ctx, "osg_layer_norm: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

// RMSNorm as the LLM graphs spell it -- Pow(x, 2) -> ReduceMean(-1) -> Add(eps) -> Sqrt -> Div(1, .) -> Mul(x, .) -> Mul(w, .) -- when the Model's
// m_requires_upcast runs that chain in fp32 (src/llm.cpp:379-383): fp32 from the f16 input to the single rounding of the result, same op order.
int osg_rms_norm(osg_ctx* ctx, osg_dtype dtype, const void* x, const void* w, void* y, long rows, int C, float eps) {
    if (rows <= 0 || C <= 0) return 0;
    if (dtype != OSG_F16) OSG_FAIL(ctx, "osg_rms_norm: only f16 storage is implemented");
    hipLaunchKernelGGL(rms_norm_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->compute, (const f16*)x, (const f16*)w, (f16*)y, C, eps);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_reduce_mean_last(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, long rows, long C) {
    if (rows <= 0 || C <= 0) return 0;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == OSG_F16)
        hipLaunchKernelGGL(reduce_mean_kernel<f16>, grid, block, 0, ctx->compute, (const f16*)x, (f16*)y, rows, C);
    else if (dtype == OSG_F32)
        hipLaunchKernelGGL(reduce_mean_kernel<float>, grid, block, 0, ctx->compute, (const float*)x, (float*)y, rows, C);
    else
        OSG_FAIL(ctx, "osg_reduce_mean_last: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_softmax_last(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, long rows, long C) {
    if (rows <= 0 || C <= 0) return 0;
    if (dtype == OSG_F16)
        hipLaunchKernelGGL(softmax_kernel<f16>, dim3((unsigned)rows), dim3(256), 0, ctx->compute, (const f16*)x, (f16*)y, C);
    else if (dtype == OSG_F32)
        hipLaunchKernelGGL(softmax_kernel<float>, dim3((unsigned)rows), dim3(256), 0, ctx->compute, (const float*)x, (float*)y, C);
    else
        OSG_FAIL(ctx, "osg_softmax_last: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
