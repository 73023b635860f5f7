// libosgpu: data-movement kernels (byte work, HBM-bound).
// Replaces XnnPack::transpose (reference onnxstream.cpp:1748), the Concat/Split/Slice/Resize/Gather host loops of
// Model::run (:4140-4299, :5999-6119, :6499-6695, :6120-6315, :6316-6498) and XnnPack::maxpool_nhwc (:1536).
#include "osg_common.h"

namespace {

constexpr int kMaxRank = 6;

struct TrParams {
    long oshape[kMaxRank];
    long istride[kMaxRank];  // input element stride for each OUTPUT dim
    int rank;
};

template <typename E>
__global__ __launch_bounds__(256) void transpose_nd_kernel(const E* __restrict__ x, E* __restrict__ y, long n, TrParams p) {
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long rem = i, off = 0;
#pragma unroll
        for (int d = kMaxRank - 1; d >= 0; d--) {
            if (d < p.rank) {
                long q = rem / p.oshape[d];
                off += (rem - q * p.oshape[d]) * p.istride[d];
                rem = q;
            }
        }
        y[i] = x[off];
    }
}

// batched 2-D transpose through LDS: x:[B, R, Cc] -> y:[B, Cc, R]   (covers NCHW<->NHWC and head split/merge)
template <typename E>
__global__ __launch_bounds__(256) void transpose_2d_kernel(const E* __restrict__ x, E* __restrict__ y, long R, long Cc) {
    __shared__ E tile[32][33];
    const long b = blockIdx.z;
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const E* xb = x + b * R * Cc;
    E* yb = y + b * R * Cc;
    for (int r = ty; r < 32; r += 8)
        if (r0 + r < R && c0 + tx < Cc) tile[r][tx] = xb[(r0 + r) * Cc + c0 + tx];
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
        if (c0 + c < Cc && r0 + tx < R) yb[(c0 + c) * R + r0 + tx] = tile[tx][c];
}

template <typename E>
__global__ __launch_bounds__(256) void copy_2d_kernel(const E* __restrict__ src, long src_pitch, long src_off, E* __restrict__ dst,
                                                      long dst_pitch, long dst_off, long outer, long inner) {
    long n = outer * inner;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long o = i / inner, k = i - o * inner;
        dst[o * dst_pitch + dst_off + k] = src[o * src_pitch + src_off + k];
    }
}

// dst[o][0:ia) = a[o][:], dst[o][ia:ia+ib) = b[o][:]  (both sources dense): the UNet's skip-connection Concat in ONE launch
template <typename E>
__global__ __launch_bounds__(256) void concat2_kernel(const E* __restrict__ a, long ia, const E* __restrict__ b, long ib, E* __restrict__ dst, long outer) {
    const long w = ia + ib, n = outer * w;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long o = i / w, k = i - o * w;
        dst[i] = k < ia ? a[o * ia + k] : b[o * ib + (k - ia)];
    }
}

template <typename E>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const E* __restrict__ x, E* __restrict__ y, int N, int C, int H, int W, int Ho,
                                                             int Wo, int nhwc, float sh_inv, float sw_inv) {
    long n = (long)N * C * Ho * Wo;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int c, wo, ho, b;
        long rem = i;
        if (nhwc) {
            c = (int)(rem % C); rem /= C;
            wo = (int)(rem % Wo); rem /= Wo;
            ho = (int)(rem % Ho); b = (int)(rem / Ho);
        } else {
            wo = (int)(rem % Wo); rem /= Wo;
            ho = (int)(rem % Ho); rem /= Ho;
            c = (int)(rem % C); b = (int)(rem / C);
        }
        // asymmetric coordinate transform + floor (reference :6120-6315): src = floor(dst / scale)
        int hi = min((int)floorf((float)ho * sh_inv), H - 1);
        int wi = min((int)floorf((float)wo * sw_inv), W - 1);
        long src = nhwc ? (((long)b * H + hi) * W + wi) * C + c : (((long)b * C + c) * H + hi) * W + wi;
        y[i] = x[src];
    }
}

template <typename E>
__global__ __launch_bounds__(256) void gather_rows_kernel(const E* __restrict__ x, const int64_t* __restrict__ idx, E* __restrict__ y,
                                                          long n_idx, long row, long n_rows) {
    long n = n_idx * row;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long r = i / row, k = i - r * row;
        int64_t s = idx[r];
        if (s < 0) s += n_rows;
        y[i] = x[s * row + k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho,
                                                           int Wo, int KH, int KW, int sh, int sw, int pt, int pl) {
    long n = (long)N * Ho * Wo * C;
    long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long rem = i;
        int c = (int)(rem % C); rem /= C;
        int wo = (int)(rem % Wo); rem /= Wo;
        int ho = (int)(rem % Ho);
        int b = (int)(rem / Ho);
        float m = -INFINITY;
        for (int kh = 0; kh < KH; kh++) {
            int hi = ho * sh - pt + kh;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int kw = 0; kw < KW; kw++) {
                int wi = wo * sw - pl + kw;
                if ((unsigned)wi >= (unsigned)W) continue;
                m = fmaxf(m, to_f32<T>(x[(((long)b * H + hi) * W + wi) * C + c]));
            }
        }
        y[i] = from_f32<T>(m);
    }
}

inline unsigned grid_for(long n) {
    long b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (unsigned)b;
}

template <typename E>
int run_transpose(osg_ctx* ctx, const E* x, E* y, int rank, const long* shape, const int* perm) {
    long n = 1;
    for (int d = 0; d < rank; d++) n *= shape[d];
    if (n == 0) return 0;
    // drop size-1 dims and merge dims that stay adjacent -> canonical (shape, perm)
    long cs[kMaxRank]; int cp[kMaxRank]; int cr = 0;
    {
        // in output order, walk input dims perm[d]; merge when perm[d] == perm[d-1]+1 (ignoring 1-sized dims)
        int in_dim_of_out[kMaxRank]; int no = 0;
        for (int d = 0; d < rank; d++)
            if (shape[perm[d]] != 1) in_dim_of_out[no++] = perm[d];
        if (no == 0) { hipMemcpyAsync(y, x, sizeof(E), hipMemcpyDeviceToDevice, ctx->compute); return 0; }
        // groups of consecutive input dims (ignoring 1-dims between them)
        auto next_nontrivial = [&](int d) { d++; while (d < rank && shape[d] == 1) d++; return d; };
        // assign group ids in output order
        int grp_first_in[kMaxRank]; long grp_size[kMaxRank]; int ng = 0;
        for (int o = 0; o < no; o++) {
            int id = in_dim_of_out[o];
            if (o > 0 && next_nontrivial(in_dim_of_out[o - 1]) == id) {
                grp_size[ng - 1] *= shape[id];
            } else {
                grp_first_in[ng] = id; grp_size[ng] = shape[id]; ng++;
            }
        }
        // canonical input order = groups sorted by first input dim
        int order[kMaxRank];
        for (int g = 0; g < ng; g++) order[g] = g;
        for (int a = 0; a < ng; a++)
            for (int b = a + 1; b < ng; b++)
                if (grp_first_in[order[b]] < grp_first_in[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
        int pos_in[kMaxRank];
        for (int k = 0; k < ng; k++) { cs[k] = grp_size[order[k]]; pos_in[order[k]] = k; }
        for (int g = 0; g < ng; g++) cp[g] = pos_in[g];
        cr = ng;
    }
    if (cr == 1) {
        OSG_HIP(ctx, hipMemcpyAsync(y, x, n * sizeof(E), hipMemcpyDeviceToDevice, ctx->compute));
        return 0;
    }
    // batched 2-D transpose?  canonical perm (0,2,1) or (1,0)
    if ((cr == 2 && cp[0] == 1 && cp[1] == 0) || (cr == 3 && cp[0] == 0 && cp[1] == 2 && cp[2] == 1)) {
        long B = cr == 3 ? cs[0] : 1, R = cs[cr - 2], Cc = cs[cr - 1];
        if (B <= 65535 && (R + 31) / 32 <= 65535) {
            dim3 grid((unsigned)((Cc + 31) / 32), (unsigned)((R + 31) / 32), (unsigned)B);
            hipLaunchKernelGGL(transpose_2d_kernel<E>, grid, dim3(256), 0, ctx->compute, x, y, R, Cc);
            OSG_LAUNCH_CHECK(ctx);
            return 0;
        }
    }
    TrParams p{};
    p.rank = cr;
    long istr[kMaxRank]; long s = 1;
    for (int d = cr - 1; d >= 0; d--) { istr[d] = s; s *= cs[d]; }
    for (int d = 0; d < cr; d++) { p.oshape[d] = cs[cp[d]]; p.istride[d] = istr[cp[d]]; }
    hipLaunchKernelGGL(transpose_nd_kernel<E>, dim3(grid_for(n)), dim3(256), 0, ctx->compute, x, y, n, p);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // namespace

extern "C" {

int osg_transpose(osg_ctx* ctx, int elem_size, const void* x, void* y, int rank, const long* shape, const int* perm) {
    if (rank < 1 || rank > kMaxRank) OSG_FAIL(ctx, "osg_transpose: rank must be in [1,6]");
    bool seen[kMaxRank] = {false};
    for (int d = 0; d < rank; d++) {
        if (perm[d] < 0 || perm[d] >= rank || seen[perm[d]]) OSG_FAIL(ctx, "osg_transpose: invalid index in perm");
        seen[perm[d]] = true;
    }
    switch (elem_size) {
        case 1: return run_transpose<uint8_t>(ctx, (const uint8_t*)x, (uint8_t*)y, rank, shape, perm);
        case 2: return run_transpose<uint16_t>(ctx, (const uint16_t*)x, (uint16_t*)y, rank, shape, perm);
        case 4: return run_transpose<uint32_t>(ctx, (const uint32_t*)x, (uint32_t*)y, rank, shape, perm);
        case 8: return run_transpose<uint64_t>(ctx, (const uint64_t*)x, (uint64_t*)y, rank, shape, perm);
    }
    OSG_FAIL(ctx, "osg_transpose: invalid element size");
}

int osg_copy_2d(osg_ctx* ctx, int elem_size, const void* src, long src_pitch, long src_off, void* dst, long dst_pitch, long dst_off,
                long outer, long inner) {
    if (outer <= 0 || inner <= 0) return 0;
    // widen the element when everything is a multiple of 16/8/4 bytes
    long es = elem_size;
    auto all_mult = [&](long m) {
        return (src_pitch * es) % m == 0 && (src_off * es) % m == 0 && (dst_pitch * es) % m == 0 && (dst_off * es) % m == 0 &&
               (inner * es) % m == 0 && ((uintptr_t)src % m) == 0 && ((uintptr_t)dst % m) == 0;
    };
    long w = all_mult(16) ? 16 : all_mult(8) ? 8 : all_mult(4) ? 4 : all_mult(2) ? 2 : 1;
    if (w < es) w = es;
    long f = w / es;
    long sp = src_pitch / f, so = src_off / f, dp = dst_pitch / f, dof = dst_off / f, in = inner / f;
    unsigned g = grid_for(outer * in);
#define OSG_CP(E) hipLaunchKernelGGL(copy_2d_kernel<E>, dim3(g), dim3(256), 0, ctx->compute, (const E*)src, sp, so, (E*)dst, dp, dof, outer, in)
    if (w == 16) OSG_CP(uint4);
    else if (w == 8) OSG_CP(uint64_t);
    else if (w == 4) OSG_CP(uint32_t);
    else if (w == 2) OSG_CP(uint16_t);
    else OSG_CP(uint8_t);
#undef OSG_CP
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_concat2(osg_ctx* ctx, int elem_size, const void* a, long inner_a, const void* b, long inner_b, void* dst, long outer) {
    if (outer <= 0 || inner_a <= 0 || inner_b <= 0) OSG_FAIL(ctx, "osg_concat2: invalid argument");
    const long es = elem_size;
    auto all_mult = [&](long m) {
        return (inner_a * es) % m == 0 && (inner_b * es) % m == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst) % m) == 0;
    };
    long w = all_mult(16) ? 16 : all_mult(8) ? 8 : all_mult(4) ? 4 : all_mult(2) ? 2 : 1;
    if (w < es) w = es;
    const long f = w / es, ia = inner_a / f, ib = inner_b / f;
    const unsigned g = grid_for(outer * (ia + ib));
#define OSG_CC(E) hipLaunchKernelGGL(concat2_kernel<E>, dim3(g), dim3(256), 0, ctx->compute, (const E*)a, ia, (const E*)b, ib, (E*)dst, outer)
    if (w == 16) OSG_CC(uint4);
    else if (w == 8) OSG_CC(uint64_t);
    else if (w == 4) OSG_CC(uint32_t);
    else if (w == 2) OSG_CC(uint16_t);
    else OSG_CC(uint8_t);
#undef OSG_CC
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_resize_nearest(osg_ctx* ctx, int elem_size, const void* x, void* y, int N, int C, int H, int W, int Ho, int Wo, int nhwc) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) OSG_FAIL(ctx, "osg_resize_nearest: invalid argument");
    long n = (long)N * C * Ho * Wo;
    float shi = (float)H / (float)Ho, swi = (float)W / (float)Wo;
    unsigned g = grid_for(n);
    if (nhwc && ((long)C * elem_size) % 16 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        // channels-last: a pixel's channel run moves as 16-byte words (the UNet's 2x upsamples: 3 launches per pass)
        const int C16 = (int)((long)C * elem_size / 16);
        g = grid_for((long)N * C16 * Ho * Wo);
        hipLaunchKernelGGL(resize_nearest_kernel<uint4>, dim3(g), dim3(256), 0, ctx->compute, (const uint4*)x, (uint4*)y, N, C16, H, W, Ho, Wo, 1, shi,
                           swi);
    } else if (elem_size == 2)
        hipLaunchKernelGGL(resize_nearest_kernel<uint16_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint16_t*)x, (uint16_t*)y, N, C, H,
                           W, Ho, Wo, nhwc, shi, swi);
    else if (elem_size == 4)
        hipLaunchKernelGGL(resize_nearest_kernel<uint32_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint32_t*)x, (uint32_t*)y, N, C, H,
                           W, Ho, Wo, nhwc, shi, swi);
    else if (elem_size == 1)
        hipLaunchKernelGGL(resize_nearest_kernel<uint8_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint8_t*)x, (uint8_t*)y, N, C, H, W,
                           Ho, Wo, nhwc, shi, swi);
    else
        OSG_FAIL(ctx, "osg_resize_nearest: invalid element size");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_gather_rows(osg_ctx* ctx, int elem_size, const void* x, const int64_t* idx, void* y, long n_idx, long row, long n_rows) {
    if (n_idx <= 0 || row <= 0) return 0;
    unsigned g = grid_for(n_idx * row);
    if (elem_size == 2)
        hipLaunchKernelGGL(gather_rows_kernel<uint16_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint16_t*)x, idx, (uint16_t*)y, n_idx,
                           row, n_rows);
    else if (elem_size == 4)
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint32_t*)x, idx, (uint32_t*)y, n_idx,
                           row, n_rows);
    else if (elem_size == 8)
        hipLaunchKernelGGL(gather_rows_kernel<uint64_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint64_t*)x, idx, (uint64_t*)y, n_idx,
                           row, n_rows);
    else if (elem_size == 1)
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3(g), dim3(256), 0, ctx->compute, (const uint8_t*)x, idx, (uint8_t*)y, n_idx, row,
                           n_rows);
    else
        OSG_FAIL(ctx, "osg_gather_rows: invalid element size");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_maxpool_nhwc(osg_ctx* ctx, osg_dtype dtype, const void* x, void* y, int N, int H, int W, int C, int KH, int KW, int sh, int sw,
                     int pt, int pl, int pb, int pr) {
    int Ho = (H + pt + pb - KH) / sh + 1, Wo = (W + pl + pr - KW) / sw + 1;
    if (Ho <= 0 || Wo <= 0) OSG_FAIL(ctx, "osg_maxpool_nhwc: empty output");
    long n = (long)N * Ho * Wo * C;
    if (dtype == OSG_F16)
        hipLaunchKernelGGL(maxpool_nhwc_kernel<f16>, dim3(grid_for(n)), dim3(256), 0, ctx->compute, (const f16*)x, (f16*)y, N, H, W, C, Ho,
                           Wo, KH, KW, sh, sw, pt, pl);
    else if (dtype == OSG_F32)
        hipLaunchKernelGGL(maxpool_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ctx->compute, (const float*)x, (float*)y, N, H, W, C,
                           Ho, Wo, KH, KW, sh, sw, pt, pl);
    else
        OSG_FAIL(ctx, "osg_maxpool_nhwc: unsupported dtype");
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
