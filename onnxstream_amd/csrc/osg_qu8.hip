// libosgpu: uint8 arithmetic -- the device side of the reference's m_use_uint8_arithmetic (W8A8) path, i.e. of the XNNPACK qu8 operators
// `class XnnPack` runs for it (reference src/onnxstream.cpp: convolution<uint8_t,int32_t> :1292 + :1458-1491, matrix_multiply<uint8_t> :1035,
// add / multiply :1666 / :846 with quint8 params, softmax<uint8_t> :1958) and of the two inline uint8 branches of Model::run (Sigmoid
// :4412-4481, InstanceNormalization :4987-5043).  The contract is BIT-EXACT codes (north_star: "bit-exact for int8 indexing"); the
// arithmetic specification is oracle/np_qu8.py, itself pinned code for code against the reference's own intermediates.
//
//   contraction: acc = sum_k (x_k - zx)(w_k - zw) on v_mfma_i32_16x16x64_i8.  The MFMA is signed x signed, the codes are unsigned: both
//     operands are re-biased by 128 on their way into LDS (x' = x ^ 0x80 = x - 128 as int8) and the identity
//         sum (x'+a)(w'+b) = sum x'w' + b sum x' + a sum w' + K a b,     a = 128 - zx,  b = 128 - zw
//     is closed in the epilogue: the row sums of x' and the column sums of w' are accumulated beside the MFMAs from the very fragments
//     they consume (v_dot4c_i32_i8 against 0x01010101).  Convolution padding is the INPUT ZERO POINT, like XNNPACK's: an out-of-image tap
//     is filled with zx, i.e. contributes exactly 0.  Then  + (int32)(bias / (sx sw))  (the reference's truncation, :4650-4656), fp32
//     requantisation  float(acc) * ((sx sw) / so), clamp to [0 - zo, 255 - zo], round to nearest even, + zo  (XNNPACK "minmax fp32").
//   elementwise: Mul = integer product + fp32 requantisation, Add = XNNPACK's 20-bit fixed-point multipliers, Sigmoid = a 256-entry table
//     the host builds with its own expf, InstanceNormalization = per-row code histogram -> 256-entry table in f64 (the reference's double
//     arithmetic on 256 distinct values) -> lookup, Softmax = XNNPACK's exp table + integer normalisation.
#include "osg_common.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

struct Q8Params {
    const uint8_t* A;        // activation codes [M, K] (lda) or NHWC image (CONV)
    const uint8_t* Bt;       // weight codes [N, K]
    uint8_t* C;              // [M, N]
    const float* bias;       // fp32 bias [N] or NULL
    int M, N, K;
    long lda;
    long strideA, strideB, strideC;
    int a_zp, b_zp, out_zp;
    float ab_scale, out_scale;   // sx * sw (float product formed on the host), so
    int H, W, Cin, Ho, Wo, KW, sh, sw, pt, pl;
    // q8_gemm2_kernel
    int KH, mt, nt;
    unsigned a_bytes, b_bytes;
    const int* wtap;         // CONV: [N][KH*KW] sums of the weight codes per filter tap (border rows: what the zero-filled halo owes), else NULL
};

__device__ __forceinline__ uint8_t q8_requant(int acc, float scale, int out_zp) {
#pragma clang fp contract(off)
    float v = (float)acc * scale;
    v = fminf(fmaxf(v, (float)(0 - out_zp)), (float)(255 - out_zp));
    return (uint8_t)((int)__builtin_rintf(v) + out_zp);
}

// The shared epilogue: acc (the MFMA's C layout with the weights on the "A" port: register r of lane l = output row m = mrow0 + 16 i + (l & 15), channel
// ncol0 + 16 j + 4 (l >> 4) + r), rs / cs = this lane's row / column sums of x' / w' after the cross-lane reduction.  Everything that depends on the channel
// alone -- the column sum (a cross-lane read), a_z * cs + K a_z b_z, the truncated bias (a load and an fp32 division) -- is formed ONCE per channel, not once
// per output: at K = 9 * 128 the per-output form cost more than the k loop (the stores to C, a char type, keep the compiler from hoisting the loads itself).
template <int TM, int TN, bool CONV, bool HALO>
__device__ __forceinline__ void q8_epilogue(const Q8Params& p, const v4i (&acc)[TM][TN], const int (&rs)[TM], const int (&cs)[TN], int mrow0, int ncol0, int lane, int zb) {
    const int az = 128 - p.a_zp, bz = 128 - p.b_zp;
    const int kab = p.K * az * bz;
    float scale;
    {
#pragma clang fp contract(off)
        scale = p.ab_scale / p.out_scale;
    }
    int ncst[TN][4];
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int csn = __shfl(cs[j], (lane >> 4) * 4 + r, 64);      // the column sum lives in the lanes whose (lane & 15) is that row of the fragment
            int c = az * csn + kab;
            const int n = min(ncol0 + j * 16 + (lane >> 4) * 4 + r, p.N - 1);
            if (p.bias) {
#pragma clang fp contract(off)
                c += (int)(p.bias[n] / p.ab_scale);      // (int32_t)(b / (x_scale * w_scale)): truncation, reference :4650-4656
            }
            ncst[j][r] = c;
        }
    uint8_t* __restrict__ C = p.C + (long)zb * p.strideC;
    const int T = p.KH * p.KW;
    const bool vec4 = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int m = mrow0 + i * 16 + (lane & 15);
        unsigned halo = 0;          // HALO (the pipelined kernel's zero-filled taps): the filter taps that fall outside the image for this output pixel
        if (HALO && p.wtap && m < p.M) {
            const int hw = p.Ho * p.Wo, n_img = m / hw, r2 = m - n_img * hw, ho = r2 / p.Wo, wo = r2 - ho * p.Wo;
            for (int kh = 0; kh < p.KH; kh++)
                for (int kw = 0; kw < p.KW; kw++) {
                    const int hi = ho * p.sh - p.pt + kh, wi = wo * p.sw - p.pl + kw;
                    if (!((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)) halo |= 1u << (kh * p.KW + kw);
                }
        }
        const int rterm = bz * rs[i];
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int nb = ncol0 + j * 16 + (lane >> 4) * 4;
            unsigned o = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int t = acc[i][j][r] + rterm + ncst[j][r];
                if (HALO && halo) {
                    const int n = min(nb + r, p.N - 1);
                    int sum = 0;
                    for (unsigned h = halo; h; h &= h - 1) sum += p.wtap[n * T + (__builtin_ctz(h))] - p.Cin * p.b_zp;
                    t += p.a_zp * sum;
                }
                o |= (unsigned)q8_requant(t, scale, p.out_zp) << (8 * r);
            }
            if (m < p.M) {
                if (vec4 && nb + 3 < p.N) *reinterpret_cast<unsigned*>(C + (long)m * p.N + nb) = o;
                else {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (nb + r < p.N) C[(long)m * p.N + nb + r] = (uint8_t)(o >> (8 * r));
                }
            }
        }
    }
}

// C[M,N] = requant( sum_k (A[m,k]-za)(Bt[n,k]-zb) + bias ).  64x64 tile, BK = 64 codes = one v_mfma_i32_16x16x64_i8 deep, 256 threads
// = 2x2 waves of 32x32; global -> registers -> LDS (row stride 80 B: conflict-free ds_read_b128), double-buffered.
// VEC: 16-byte chunks (K % 16 == 0, rows 16-byte aligned; CONV: Cin % 16 == 0 so a chunk never straddles a filter tap).
// BM x BN in {64 x 64, 128 x 128}: the large tile moves half the bytes per MAC (the 512x512 convolutions of the VAE decoder: 8 192 tiles of 64 x 64
// were fill-bound at 12 % of the int8 MFMA peak); a thread stages BM/64 + BN/64 chunks per k-tile, a wave owns (BM/2) x (BN/2).
template <bool CONV, bool VEC, int BM, int BN>
__global__ __launch_bounds__(256) void q8_gemm_kernel(Q8Params p) {
    constexpr int BK = 64, LDSW = BK + 16;
    constexpr int A_IT = BM / 64, B_IT = BN / 64, TM = BM / 32, TN = BN / 32;
    __shared__ __attribute__((aligned(16))) uint8_t smem[2][(BM + BN) * LDSW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, zb = blockIdx.z;
    const uint8_t* __restrict__ A = p.A + (long)zb * p.strideA;
    const uint8_t* __restrict__ Bt = p.Bt + (long)zb * p.strideB;
    const int row = tid >> 2, kc = (tid & 3) * 16;       // this thread stages 16-byte chunks of rows row, row + 64, ... of A and of B per k-tile

    // A row state
    bool a_ok[A_IT];
    long a_off[A_IT];
    int hi0[A_IT], wi0[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; it++) {
        const int am = m0 + row + it * 64;
        a_ok[it] = am < p.M;
        hi0[it] = wi0[it] = 0;
        if (CONV) {
            const int mm = a_ok[it] ? am : 0, hw = p.Ho * p.Wo, n_img = mm / hw, rem = mm - n_img * hw, ho = rem / p.Wo, wo = rem - ho * p.Wo;
            a_off[it] = (long)n_img * p.H * p.W * p.Cin;
            hi0[it] = ho * p.sh - p.pt;
            wi0[it] = wo * p.sw - p.pl;
        } else
            a_off[it] = (long)(a_ok[it] ? am : 0) * p.lda;
    }
    bool b_ok[B_IT];
    long b_off[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; it++) {
        const int bn = n0 + row + it * 64;
        b_ok[it] = bn < p.N;
        b_off[it] = (long)(b_ok[it] ? bn : 0) * p.K;
    }
    const unsigned pad4 = ((unsigned)(p.a_zp ^ 0x80) & 0xffu) * 0x01010101u;   // a padding tap holds the input zero point (re-biased)

    v4i areg[A_IT], breg[B_IT];
    auto load_tile = [&](int k0) {
        const int k = k0 + kc;
        if (VEC) {
            int kh = 0, kw = 0, c = k;
            if (CONV && k < p.K) {
                const int cell = k / p.Cin;
                c = k - cell * p.Cin;
                kh = cell / p.KW;
                kw = cell - kh * p.KW;
            }
#pragma unroll
            for (int it = 0; it < A_IT; it++) {
                v4i va = {0, 0, 0, 0};
                if (k < p.K && a_ok[it]) {
                    if (CONV) {
                        const int hi = hi0[it] + kh, wi = wi0[it] + kw;
                        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                            va = *reinterpret_cast<const v4i*>(A + a_off[it] + ((long)hi * p.W + wi) * p.Cin + c);
                            va ^= (int)0x80808080;
                        } else
                            va = v4i{(int)pad4, (int)pad4, (int)pad4, (int)pad4};
                    } else {
                        va = *reinterpret_cast<const v4i*>(A + a_off[it] + k);
                        va ^= (int)0x80808080;
                    }
                }
                areg[it] = va;
            }
#pragma unroll
            for (int it = 0; it < B_IT; it++) {
                v4i vb = {0, 0, 0, 0};
                if (k < p.K && b_ok[it]) {
                    vb = *reinterpret_cast<const v4i*>(Bt + b_off[it] + k);
                    vb ^= (int)0x80808080;
                }
                breg[it] = vb;
            }
        } else {
#pragma unroll
            for (int it = 0; it < A_IT; it++) {
                unsigned wa[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int ke = k + e;
                    unsigned xa = 0;     // beyond K: x' = 0 (no contribution to any term)
                    if (ke < p.K && a_ok[it]) {
                        if (CONV) {
                            const int cell = ke / p.Cin, c = ke - cell * p.Cin, kh = cell / p.KW, kw = cell - kh * p.KW;
                            const int hi = hi0[it] + kh, wi = wi0[it] + kw;
                            const unsigned code = ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                                                      ? A[a_off[it] + ((long)hi * p.W + wi) * p.Cin + c] : (unsigned)p.a_zp;
                            xa = (code ^ 0x80u) & 0xffu;
                        } else
                            xa = ((unsigned)A[a_off[it] + ke] ^ 0x80u) & 0xffu;
                    }
                    wa[e >> 2] |= xa << ((e & 3) * 8);
                }
                areg[it] = v4i{(int)wa[0], (int)wa[1], (int)wa[2], (int)wa[3]};
            }
#pragma unroll
            for (int it = 0; it < B_IT; it++) {
                unsigned wb[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int ke = k + e;
                    unsigned xb = 0;
                    if (ke < p.K && b_ok[it]) xb = ((unsigned)Bt[b_off[it] + ke] ^ 0x80u) & 0xffu;
                    wb[e >> 2] |= xb << ((e & 3) * 8);
                }
                breg[it] = v4i{(int)wb[0], (int)wb[1], (int)wb[2], (int)wb[3]};
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; it++) *reinterpret_cast<v4i*>(&smem[buf][(row + it * 64) * LDSW + kc]) = areg[it];
#pragma unroll
        for (int it = 0; it < B_IT; it++) *reinterpret_cast<v4i*>(&smem[buf][(BM + row + it * 64) * LDSW + kc]) = breg[it];
    };

    v4i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = v4i{0, 0, 0, 0};
    int rs[TM], cs[TN];     // partial sums of x' over this lane's k-chunks of its activation rows / of w' of its weight rows
#pragma unroll
    for (int i = 0; i < TM; i++) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < TN; j++) cs[j] = 0;

    const int nkt = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int frow = lane & 15, fk = (lane >> 4) * 16;
    for (int kt = 0; kt < nkt; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile((kt + 1) * BK);
        v4i xa[TM], wb[TN];
#pragma unroll
        for (int i = 0; i < TM; i++) xa[i] = *reinterpret_cast<const v4i*>(&smem[cur][(wm0 + i * 16 + frow) * LDSW + fk]);
#pragma unroll
        for (int j = 0; j < TN; j++) wb[j] = *reinterpret_cast<const v4i*>(&smem[cur][(BM + wn0 + j * 16 + frow) * LDSW + fk]);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wb[j], xa[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) rs[i] = __builtin_amdgcn_sdot4(xa[i][e], 0x01010101, rs[i], false);
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) cs[j] = __builtin_amdgcn_sdot4(wb[j][e], 0x01010101, cs[j], false);
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }
    // the four 16-lane groups hold different k-chunks of the same rows
#pragma unroll
    for (int i = 0; i < TM; i++) { rs[i] += __shfl_xor(rs[i], 16, 64); rs[i] += __shfl_xor(rs[i], 32, 64); }
#pragma unroll
    for (int j = 0; j < TN; j++) { cs[j] += __shfl_xor(cs[j], 16, 64); cs[j] += __shfl_xor(cs[j], 32, 64); }
    q8_epilogue<TM, TN, CONV, false>(p, acc, rs, cs, m0 + wm0, n0 + wn0, lane, zb);
}

// =====================================================================================================================================================
// v2 (round 3): the direct-to-LDS pipelined form, built like gemm2_kernel of osg_gemm.hip -- both code matrices stream HBM/L2 -> LDS with
// `buffer_load_dwordx4 ... lds` into an NST-deep ring of [BM + BN][128 B] stages (chunk c of row r at slot c ^ (r & 7): conflict-free ds_read_b128), counted
// s_waitcnt vmcnt(N), ONE s_barrier per 128-code k-tile = two v_mfma_i32_16x16x64_i8 steps, XCD-aware walk of a 1-D grid.  What differs from the f16 kernel:
//   * the DMA cannot re-bias the codes on their way in: the fragments are xor'ed with 0x80 AFTER the ds_read (4 VALU per fragment, in the MFMAs' shadow);
//     the row sums of x' and the column sums of w' that close  sum (x'+a)(w'+b)  come from the same fragments (v_dot4 against 0x01010101), as before;
//   * the descriptor's bounds check zero-fills the convolution halo with CODE 0, XNNPACK pads with the input zero point (a tap that contributes exactly 0):
//     every out-of-image tap therefore owes  zx * sum_c (w[n,tap,c] - zw)  -- added in the epilogue of the (few) border rows from a per-weight table of
//     tap sums wtap[N][KH*KW] (built once per weight, q8_tap_sums_kernel).  Integer arithmetic: the codes are those of the v1 kernel, bit for bit.
// Requirements: K % 128 == 0 (CONV: Cin % 128 == 0), 16-byte aligned rows, N % 4 == 0, KH * KW <= 32.
// WGM x 2 waves (WGM = 2: 256 threads; 4: 512 threads = two waves per SIMD inside one workgroup: one wave's fragment reads and re-biasing run under the
// other's MFMAs).  DBG (timing experiments only, wrong results): 1 = no column sums / no re-biasing of the weights, 2 = no row sums / re-biasing either.
template <int BM, int BN, int NST, bool CONV, int WGM = 2, int DBG = 0>
__global__ __launch_bounds__(WGM * 128) void q8_gemm2_kernel(Q8Params p) {
    constexpr int ROWB = 128, NW = WGM * 2;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LD = BM / (8 * NW), B_LD = BN / (8 * NW);   // 1-KiB wave-loads per wave per k-tile
    constexpr int WM = BM / WGM, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int INFLIGHT = (NST - 2) * (A_LD + B_LD);
    constexpr unsigned OOB = 0x80000000u;
    static_assert(INFLIGHT <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smemq[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    int L;
    {
        const int total = gridDim.x, bid = blockIdx.x, x = bid & 7, i = bid >> 3, q = total >> 3, r = total & 7;
        L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int per_batch = p.mt * p.nt;
    const int zb = L / per_batch, rem = L - zb * per_batch;
    const int m_tile = rem / p.nt, n_tile = rem - m_tile * p.nt;      // n inner: neighbours in the walk share the activation tile through their XCD's L2
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int nkt = p.K >> 7;

    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)zb * p.strideA), 0, p.a_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Bt + (long)zb * p.strideB), 0, p.b_bytes, 0x00020000);

    const int rsub = lane >> 3, gch = (lane & 7) ^ rsub;
    int a_base[A_LD], a_hi0[A_LD], a_wi0[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; j++) {
        const int m = m0 + (j * NW + wave) * 8 + rsub;
        if (CONV) {
            const int mm = m < p.M ? m : 0, hw = p.Ho * p.Wo, n_img = mm / hw, r2 = mm - n_img * hw, ho = r2 / p.Wo, wo = r2 - ho * p.Wo;
            a_hi0[j] = m < p.M ? ho * p.sh - p.pt : -0x40000000;
            a_wi0[j] = wo * p.sw - p.pl;
            a_base[j] = ((n_img * p.H + (ho * p.sh - p.pt)) * p.W + (wo * p.sw - p.pl)) * p.Cin + gch * 16;
        } else {
            a_base[j] = m < p.M ? (int)((long)m * p.lda + gch * 16) : (int)OOB;
            a_hi0[j] = a_wi0[j] = 0;
        }
    }
    int b_base[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; j++) {
        const int n = n0 + (j * NW + wave) * 8 + rsub;
        b_base[j] = n < p.N ? (int)((long)n * p.K + gch * 16) : (int)OOB;
    }
    int ik = 0, i_c0 = 0, i_kh = 0, i_kw = 0;
    auto issue_tile = [&](int stage) {
        char* As = smemq + stage * STAGE;
        char* Bs = As + A_BYTES;
        const bool live = ik < p.K;
        const unsigned kill = live ? 0u : OOB;      // past the last k-tile: dummy (zero-filling) loads keep vmcnt uniform
        if (CONV) {
            const int tap_off = (i_kh * p.W + i_kw) * p.Cin + i_c0;
#pragma unroll
            for (int j = 0; j < A_LD; j++) {
                const int hi = a_hi0[j] + i_kh, wi = a_wi0[j] + i_kw;
                const bool ok = live && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(As + (j * NW + wave) * 1024), 16, ok ? (unsigned)(a_base[j] + tap_off) : OOB, 0, 0, 0);
            }
            i_c0 += 128;
            if (i_c0 >= p.Cin) { i_c0 = 0; if (++i_kw == p.KW) { i_kw = 0; ++i_kh; } }
        } else {
#pragma unroll
            for (int j = 0; j < A_LD; j++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(As + (j * NW + wave) * 1024), 16, (unsigned)a_base[j] | kill, ik, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_LD; j++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(Bs + (j * NW + wave) * 1024), 16, (unsigned)b_base[j] | kill, ik, 0, 0);
        ik += 128;
    };

    v4i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = v4i{0, 0, 0, 0};
    int rs[TM], cs[TN];
#pragma unroll
    for (int i = 0; i < TM; i++) rs[i] = 0;
#pragma unroll
    for (int j = 0; j < TN; j++) cs[j] = 0;

    const int frow = lane & 15;
    const int fsw = (((lane >> 4) ^ (frow & 7)) << 4);
    const int a_rd = (wm0 + frow) * ROWB + fsw;
    const int b_rd = A_BYTES + (wn0 + frow) * ROWB + fsw;

#pragma unroll
    for (int s2 = 0; s2 < NST - 1; s2++) issue_tile(s2);
    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < nkt; kt++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");   // my share of tile kt has landed
        __builtin_amdgcn_s_barrier();                                      // everyone's has; tile kt-1's buffer is free
        issue_tile(nxt);
        const char* St = smemq + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            v4i xa[TM], wb[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                xa[i] = *reinterpret_cast<const v4i*>(St + ((a_rd + i * 16 * ROWB) ^ (ks << 6)));
                if (DBG < 2) xa[i] ^= (int)0x80808080;
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                wb[j] = *reinterpret_cast<const v4i*>(St + ((b_rd + j * 16 * ROWB) ^ (ks << 6)));
                if (DBG < 1) wb[j] ^= (int)0x80808080;
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wb[j], xa[i], acc[i][j], 0, 0, 0);
            if (DBG < 2) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int e = 0; e < 4; e++) rs[i] = __builtin_amdgcn_sdot4(xa[i][e], 0x01010101, rs[i], false);
            }
            if (DBG < 1) {
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cs[j] = __builtin_amdgcn_sdot4(wb[j][e], 0x01010101, cs[j], false);
            }
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the dummy tail loads before the LDS is released

#pragma unroll
    for (int i = 0; i < TM; i++) { rs[i] += __shfl_xor(rs[i], 16, 64); rs[i] += __shfl_xor(rs[i], 32, 64); }
#pragma unroll
    for (int j = 0; j < TN; j++) { cs[j] += __shfl_xor(cs[j], 16, 64); cs[j] += __shfl_xor(cs[j], 32, 64); }
    q8_epilogue<TM, TN, CONV, CONV>(p, acc, rs, cs, m0 + wm0, n0 + wn0, lane, zb);
}

// wtap[n][tap] = sum_c w[n][tap][c] (codes as they are), one wave per (n, tap)
__global__ __launch_bounds__(64) void q8_tap_sums_kernel(const uint8_t* __restrict__ w, int* __restrict__ wtap, int Cin) {
    const uint8_t* __restrict__ src = w + (long)blockIdx.x * Cin;
    int s = 0;
    for (int c = threadIdx.x; c < Cin; c += 64) s += src[c];
    for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d, 64);
    if (threadIdx.x == 0) wtap[blockIdx.x] = s;
}

template <int BM, int BN, int NST, bool CONV, int WGM = 2, int DBG = 0>
int launch_q8v2(osg_ctx* ctx, Q8Params& p, int batch) {
    constexpr size_t smem = (size_t)NST * (BM + BN) * 128;
    static_assert(smem <= 160 * 1024, "LDS budget");
    auto kern = q8_gemm2_kernel<BM, BN, NST, CONV, WGM, DBG>;
    static unsigned long long attr_mask = 0;   // (per device: hipFuncSetAttribute is, and a process may hold several)
    if (osg_first_on_device(attr_mask)) {
        OSG_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    }
    p.mt = (p.M + BM - 1) / BM;
    p.nt = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.mt * p.nt * batch)), dim3(WGM * 128), smem, ctx->compute, p);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

// the pipelined kernel where it applies (OSG_QU8_V2=0: never, 2: whenever the shape is legal; OSG_QU8_NST=2|3|4 pins the ring depth -- for tests and A/B runs)
static int try_q8v2(osg_ctx* ctx, Q8Params p, int batch, bool conv, bool* taken) {
    *taken = false;
    const int v2 = getenv("OSG_QU8_V2") ? atoi(getenv("OSG_QU8_V2")) : 1;       // (read per launch: the tests switch it inside one process.  2: also for small problems)
    const int nst_env = getenv("OSG_QU8_NST") ? atoi(getenv("OSG_QU8_NST")) : 0;
    if (!v2) return 0;
    const bool al = ((((uintptr_t)p.A | (uintptr_t)p.Bt | (uintptr_t)p.C) & 15) == 0) && p.strideA % 16 == 0 && p.strideB % 16 == 0 && p.strideC % 4 == 0;
    const int taps = conv ? p.KH * p.KW : 1;
    if (!al || p.K % 128 || (conv ? p.Cin % 128 != 0 : p.lda % 16 != 0) || p.N % 4 || taps > 32) return 0;
    const long a_bytes = conv ? (long)(p.M / (p.Ho * p.Wo)) * p.H * p.W * p.Cin : (long)p.M * p.lda, b_bytes = (long)p.N * p.K;
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return 0;
    // which form: 128 x 128 tiles (two workgroups per CU on a 2-deep ring) where the grid still gives every CU two of them and the k extent is long enough to
    // pay for the ring's fill (at K = 9 * 128 the register-staged kernel with four co-resident workgroups is ahead: profiles/r03_q8_conv_probe.txt); 64 x 64
    // tiles on a 4-deep ring for the small images (64 x 64 x 512: 128 large tiles for 256 CUs, and the register-staged kernel exposes one global-load
    // latency per 64-code k-tile -- a chain of 72 of them per workgroup)
    const long tiles128 = (long)((p.N + 127) / 128) * ((p.M + 127) / 128) * batch;
    const long tiles64 = (long)((p.N + 63) / 64) * ((p.M + 63) / 64) * batch;
    const int tile_env = getenv("OSG_QU8_V2_TILE") ? atoi(getenv("OSG_QU8_V2_TILE")) : 0;
    int tile = 0;
    if (tiles128 >= 2L * ctx->num_cu) tile = (v2 == 2 || p.K >= 2048) ? 128 : 0;
    else if (p.N >= 64 && (v2 == 2 || (tiles64 >= ctx->num_cu / 2 && p.K >= 1024))) tile = 64;
    if (tile_env == 64 || tile_env == 128) tile = tile_env;
    if (!tile) return 0;
    p.a_bytes = (unsigned)a_bytes;
    p.b_bytes = (unsigned)b_bytes;
    if (conv && (p.pt || p.pl || (p.Ho - 1) * p.sh - p.pt + p.KH > p.H || (p.Wo - 1) * p.sw - p.pl + p.KW > p.W)) {
        if (!p.wtap) {      // no table from the caller: into the workspace, on every call (stream-ordered before the launch that reads it)
            if (osg_ensure_workspace(ctx, (size_t)p.N * taps * sizeof(int))) return 1;
            hipLaunchKernelGGL(q8_tap_sums_kernel, dim3((unsigned)(p.N * taps)), dim3(64), 0, ctx->compute, p.Bt, (int*)ctx->ws, p.Cin);
            OSG_LAUNCH_CHECK(ctx);
            p.wtap = (const int*)ctx->ws;
        }
    } else
        p.wtap = nullptr;
    *taken = true;
    const int dbg = getenv("OSG_QU8_DBG") ? atoi(getenv("OSG_QU8_DBG")) : 0;            // timing experiments (wrong results): see q8_gemm2_kernel
    const int wgm = getenv("OSG_QU8_WGM") ? atoi(getenv("OSG_QU8_WGM")) : 2;            // 4: the 256 x 128 tile on eight waves
    if (tile == 64) {
        const int nst = nst_env ? nst_env : 4;
        if (conv) return nst == 2 ? launch_q8v2<64, 64, 2, true>(ctx, p, batch) : nst == 3 ? launch_q8v2<64, 64, 3, true>(ctx, p, batch) : launch_q8v2<64, 64, 4, true>(ctx, p, batch);
        return nst == 2 ? launch_q8v2<64, 64, 2, false>(ctx, p, batch) : nst == 3 ? launch_q8v2<64, 64, 3, false>(ctx, p, batch) : launch_q8v2<64, 64, 4, false>(ctx, p, batch);
    }
    const int nst = nst_env ? nst_env : 2;
    if (conv) {
        if (dbg == 1) return nst == 2 ? launch_q8v2<128, 128, 2, true, 2, 1>(ctx, p, batch) : launch_q8v2<128, 128, 3, true, 2, 1>(ctx, p, batch);
        if (dbg == 2) return nst == 2 ? launch_q8v2<128, 128, 2, true, 2, 2>(ctx, p, batch) : launch_q8v2<128, 128, 3, true, 2, 2>(ctx, p, batch);
        if (wgm == 4) return nst == 2 ? launch_q8v2<256, 128, 2, true, 4>(ctx, p, batch) : launch_q8v2<256, 128, 3, true, 4>(ctx, p, batch);
        if (nst == 2) return launch_q8v2<128, 128, 2, true>(ctx, p, batch);
        if (nst == 4) return launch_q8v2<128, 128, 4, true>(ctx, p, batch);
        return launch_q8v2<128, 128, 3, true>(ctx, p, batch);
    }
    if (wgm == 4) return nst == 2 ? launch_q8v2<256, 128, 2, false, 4>(ctx, p, batch) : launch_q8v2<256, 128, 3, false, 4>(ctx, p, batch);
    if (nst == 2) return launch_q8v2<128, 128, 2, false>(ctx, p, batch);
    if (nst == 4) return launch_q8v2<128, 128, 4, false>(ctx, p, batch);
    return launch_q8v2<128, 128, 3, false>(ctx, p, batch);
}

int launch_q8(osg_ctx* ctx, const Q8Params& p, int batch, bool conv) {
    {
        bool taken = false;
        if (try_q8v2(ctx, p, batch, conv, &taken)) return 1;
        if (taken) return 0;
    }
    const bool al = ((((uintptr_t)p.A | (uintptr_t)p.Bt) & 15) == 0) && p.strideA % 16 == 0 && p.strideB % 16 == 0;
    const bool vec = al && p.K % 16 == 0 && (conv ? p.Cin % 16 == 0 : p.lda % 16 == 0);
    // 128 x 128 tiles where they still leave every CU at least two of them (OSG_QU8_TILE=64|128 pins the choice for tests)
    const int force_tile = getenv("OSG_QU8_TILE") ? atoi(getenv("OSG_QU8_TILE")) : 0;
    const long tiles128 = (long)((p.N + 127) / 128) * ((p.M + 127) / 128) * batch;
    const bool big = vec && force_tile != 64 && (force_tile == 128 || (p.N >= 128 && tiles128 >= 2L * ctx->num_cu));
    if (big) {
        dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, batch);
        if (grid.y > 65535u) OSG_FAIL(ctx, "osg_qu8: M too large for one launch");
        if (conv) hipLaunchKernelGGL((q8_gemm_kernel<true, true, 128, 128>), grid, dim3(256), 0, ctx->compute, p);
        else hipLaunchKernelGGL((q8_gemm_kernel<false, true, 128, 128>), grid, dim3(256), 0, ctx->compute, p);
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, batch);
    if (grid.y > 65535u) OSG_FAIL(ctx, "osg_qu8: M too large for one launch");
    if (conv) {
        if (vec) hipLaunchKernelGGL((q8_gemm_kernel<true, true, 64, 64>), grid, dim3(256), 0, ctx->compute, p);
        else hipLaunchKernelGGL((q8_gemm_kernel<true, false, 64, 64>), grid, dim3(256), 0, ctx->compute, p);
    } else {
        if (vec) hipLaunchKernelGGL((q8_gemm_kernel<false, true, 64, 64>), grid, dim3(256), 0, ctx->compute, p);
        else hipLaunchKernelGGL((q8_gemm_kernel<false, false, 64, 64>), grid, dim3(256), 0, ctx->compute, p);
    }
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

// ---- elementwise ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void q8_lut_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, long n, const uint8_t* __restrict__ lut) {
    __shared__ uint8_t t[256];
    t[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = t[x[i]];
}

constexpr int kMaxRank = 6;
struct Q8Bcast {
    long oshape[kMaxRank], astride[kMaxRank], bstride[kMaxRank];
    int rank;
};
struct Q8Bin {
    int a_zp, b_zp, out_zp;
    float scale;                 // MUL: (sa * sb) / so
    int a_mult, b_mult, bias;    // ADD: XNNPACK fixed point
    unsigned shift;
};

template <int KIND>   // 0 add, 1 mul
__global__ __launch_bounds__(256) void q8_binary_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ y, long n,
                                                        Q8Bcast p, Q8Bin q) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        long rem = i, ao = 0, bo = 0;
#pragma unroll
        for (int d = kMaxRank - 1; d >= 0; d--)
            if (d < p.rank) {
                const long qd = rem / p.oshape[d], idx = rem - qd * p.oshape[d];
                rem = qd;
                ao += idx * p.astride[d];
                bo += idx * p.bstride[d];
            }
        const int av = a[ao], bv = b[bo];
        if (KIND == 1) {
            y[i] = q8_requant((av - q.a_zp) * (bv - q.b_zp), q.scale, q.out_zp);
        } else {
            const int acc = q.bias + av * q.a_mult + bv * q.b_mult;
            const int o = (acc >> q.shift) + q.out_zp;
            y[i] = (uint8_t)min(max(o, 0), 255);
        }
    }
}

template <int KIND>
__device__ __forceinline__ uint8_t q8_bin_op(int av, int bv, const Q8Bin& q) {
    if (KIND == 1) return q8_requant((av - q.a_zp) * (bv - q.b_zp), q.scale, q.out_zp);
    const int acc = q.bias + av * q.a_mult + bv * q.b_mult;
    const int o = (acc >> q.shift) + q.out_zp;
    return (uint8_t)min(max(o, 0), 255);
}

// the two shapes the uint8 graphs are made of: a full tensor combined with (MODE 0) a tensor of the same shape or (MODE 1) an operand that
// repeats with a period -- b[(i / inner) % period]: a scalar (period 1), a per-channel vector against NHWC (inner 1) or against NCHW
// (inner H*W).  16 codes per thread, 16-byte loads and stores.
template <int KIND, int MODE>
__global__ __launch_bounds__(256) void q8_binary_fast_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ y, long n,
                                                             long inner, long period, Q8Bin q) {
    const long nv = n >> 4, stride = (long)gridDim.x * 256;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += stride) {
        const long i0 = v << 4;
        const v4i av = *reinterpret_cast<const v4i*>(a + i0);
        v4i bv;
        if (MODE == 0) bv = *reinterpret_cast<const v4i*>(b + i0);
        v4i ov;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned o = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ac = ((unsigned)av[w] >> (8 * e)) & 0xff;
                int bc;
                if (MODE == 0) bc = ((unsigned)bv[w] >> (8 * e)) & 0xff;
                else bc = b[((i0 + w * 4 + e) / inner) % period];
                o |= (unsigned)q8_bin_op<KIND>(ac, bc, q) << (8 * e);
            }
            ov[w] = (int)o;
        }
        *reinterpret_cast<v4i*>(y + i0) = ov;
    }
    // tail
    for (long i = (nv << 4) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        y[i] = q8_bin_op<KIND>(a[i], MODE == 0 ? b[i] : b[(i / inner) % period], q);
}

// NORM: the codes first go through the per-group table of a uint8 InstanceNormalization (in_lut [G][256], group = channel / cpg) -- the lookup the
// separate apply launch would have done -- then through the affine chain
template <bool ACT, bool NORM>
__global__ __launch_bounds__(256) void q8_affine_act_kernel(const uint8_t* __restrict__ x, const uint8_t* __restrict__ g, const uint8_t* __restrict__ b,
                                                            const uint8_t* __restrict__ lut_g, uint8_t* __restrict__ y, long n, int C, long inner, Q8Bin qm, Q8Bin qa,
                                                            Q8Bin qs, const uint8_t* __restrict__ in_lut, int cpg, int G) {
    __shared__ uint8_t lut[256];
    extern __shared__ uint8_t nl[];            // NORM: [G][256]
    if (ACT) lut[threadIdx.x] = lut_g[threadIdx.x];
    if (NORM)
        for (int k = threadIdx.x; k < G * 256; k += 256) nl[k] = in_lut[k];
    if (ACT || NORM) __syncthreads();
    auto one = [&](unsigned code, long i) -> unsigned {
        const int c = (int)((i / inner) % C);
        if (NORM) code = nl[(c / cpg) * 256 + code];
        const unsigned v1 = q8_bin_op<1>((int)code, g[c], qm);
        const unsigned v2 = q8_bin_op<0>((int)v1, b[c], qa);
        if (!ACT) return v2;
        return q8_bin_op<1>((int)v2, lut[v2], qs);
    };
    const long nv = n >> 4, stride = (long)gridDim.x * 256;
    const bool al = ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    if (al) {
        for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += stride) {
            const long i0 = v << 4;
            const v4i xv = *reinterpret_cast<const v4i*>(x + i0);
            v4i ov;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                unsigned o = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) o |= one(((unsigned)xv[w] >> (8 * e)) & 0xff, i0 + w * 4 + e) << (8 * e);
                ov[w] = (int)o;
            }
            *reinterpret_cast<v4i*>(y + i0) = ov;
        }
        for (long i = (nv << 4) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = (uint8_t)one(x[i], i);
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = (uint8_t)one(x[i], i);
    }
}

__device__ __forceinline__ uint8_t q8_quantize(float x, float inv_scale, int zp) {
#pragma clang fp contract(off)
    float r = __builtin_rintf(x * inv_scale) + (float)zp;
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    return (uint8_t)r;
}

// InstanceNormalization on [rows, L] codes in three launches that all fill the chip (a full-size VAE has 32 rows of a million codes):
//   1. histogram of every row's codes (workgroups own 64 KiB pieces of a row: per-wave LDS histograms -> one global atomic per bin);
//   2. per row: mean / variance from the histogram in f64 exactly as the reference accumulates them over the elements (the dequantised
//      values are 256 distinct floats; count x value products and their sums stay below 2^53 ulps, so the order of the additions cannot
//      matter), then the output code of every input code -> a 256-entry table;
//   3. lookup.
constexpr int kInPiece = 65536;
__global__ __launch_bounds__(256) void q8_in_hist_kernel(const uint8_t* __restrict__ x, unsigned* __restrict__ hist, long L) {
    __shared__ unsigned h[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, row = blockIdx.y;
    for (int k = tid; k < 1024; k += 256) (&h[0][0])[k] = 0;
    __syncthreads();
    const uint8_t* __restrict__ xr = x + (long)row * L;
    const long beg = (long)blockIdx.x * kInPiece, end = min(L, beg + kInPiece);
    if (((uintptr_t)(xr + beg) & 15) == 0) {
        const long nv = (end - beg) >> 4;
        for (long v = tid; v < nv; v += 256) {
            const v4i c = *reinterpret_cast<const v4i*>(xr + beg + (v << 4));
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int e = 0; e < 4; e++) atomicAdd(&h[wave][((unsigned)c[w] >> (8 * e)) & 0xff], 1u);
        }
        for (long i = beg + (nv << 4) + tid; i < end; i += 256) atomicAdd(&h[wave][xr[i]], 1u);
    } else {
        for (long i = beg + tid; i < end; i += 256) atomicAdd(&h[wave][xr[i]], 1u);
    }
    __syncthreads();
    const unsigned t = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    if (t) atomicAdd(&hist[row * 256 + tid], t);
}

__global__ __launch_bounds__(256) void q8_in_lut_kernel(const unsigned* __restrict__ hist_g, uint8_t* __restrict__ lut_g, long L, int n_scale,
                                                        const float* __restrict__ scale, const float* __restrict__ bias, float eps, float in_scale, int in_zp,
                                                        float out_scale, int out_zp) {
#pragma clang fp contract(off)
    __shared__ unsigned hist[256];
    __shared__ float deq[256];
    __shared__ double stat[2];
    const int row = blockIdx.x, c = threadIdx.x;
    hist[c] = hist_g[row * 256 + c];
    deq[c] = (float)(c - in_zp) * in_scale;
    __syncthreads();
    if (c == 0) {
        double mean = 0;
        for (int k = 0; k < 256; k++) mean += (double)hist[k] * (double)deq[k];
        mean /= (double)L;
        double var = 0;
        for (int k = 0; k < 256; k++) {
            const float dev = (float)((double)deq[k] - mean);
            var += (double)hist[k] * (double)(dev * dev);
        }
        var /= (double)L;
        stat[0] = mean;
        stat[1] = sqrt(var + (double)eps);
    }
    __syncthreads();
    const double sc = (double)scale[row % n_scale], bi = (double)bias[row % n_scale];
    const float v = (float)(sc * ((double)deq[c] - stat[0]) / stat[1] + bi);
    const float inv = 1.0f / out_scale;
    lut_g[row * 256 + c] = q8_quantize(v, inv, out_zp);
}

__global__ __launch_bounds__(256) void q8_in_apply_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, const uint8_t* __restrict__ lut_g, long L) {
    __shared__ uint8_t lut[256];
    const int tid = threadIdx.x, row = blockIdx.y;
    lut[tid] = lut_g[row * 256 + tid];
    __syncthreads();
    const uint8_t* __restrict__ xr = x + (long)row * L;
    uint8_t* __restrict__ yr = y + (long)row * L;
    const long beg = (long)blockIdx.x * kInPiece, end = min(L, beg + kInPiece);
    if ((((uintptr_t)(xr + beg) | (uintptr_t)(yr + beg)) & 15) == 0) {
        const long nv = (end - beg) >> 4;
        for (long v = tid; v < nv; v += 256) {
            const v4i c = *reinterpret_cast<const v4i*>(xr + beg + (v << 4));
            v4i o;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                unsigned r = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) r |= (unsigned)lut[((unsigned)c[w] >> (8 * e)) & 0xff] << (8 * e);
                o[w] = (int)r;
            }
            *reinterpret_cast<v4i*>(yr + beg + (v << 4)) = o;
        }
        for (long i = beg + (nv << 4) + tid; i < end; i += 256) yr[i] = lut[xr[i]];
    } else {
        for (long i = beg + tid; i < end; i += 256) yr[i] = lut[xr[i]];
    }
}

// The same InstanceNormalization on [1,G,L] when the tensor behind the Reshape lives as NHWC [HW][C] (row g = channels [g*cpg, (g+1)*cpg) of every
// pixel): histogram and table lookup address the codes where they lie -- no NCHW copy before, none after.  Same tables, same codes.
// sh >= 0: cpg == 1 << sh (a 16-code vector starts at a channel that is a multiple of 16, so its codes' groups are (c0 >> sh) + (j >> sh): no division)
__global__ __launch_bounds__(256) void q8_in_hist_nhwc_kernel(const uint8_t* __restrict__ x, unsigned* __restrict__ hist, long n, int C, int cpg, int G, int sh) {
    extern __shared__ unsigned hg[];          // [G][256]
    const int tid = threadIdx.x;
    for (int k = tid; k < G * 256; k += 256) hg[k] = 0;
    __syncthreads();
    const long beg = (long)blockIdx.x * kInPiece, end = min(n, beg + kInPiece);
    if ((C & 15) == 0 && ((uintptr_t)(x + beg) & 15) == 0) {
        const long nv = (end - beg) >> 4;
        for (long v = tid; v < nv; v += 256) {
            const long i0 = beg + (v << 4);
            const int c0 = (int)(i0 % C);
            const v4i c = *reinterpret_cast<const v4i*>(x + i0);
            if (sh >= 0) {
                const int gb = c0 >> sh;
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int e = 0; e < 4; e++) atomicAdd(&hg[(gb + ((w * 4 + e) >> sh)) * 256 + (((unsigned)c[w] >> (8 * e)) & 0xff)], 1u);
            } else {
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int e = 0; e < 4; e++) atomicAdd(&hg[((c0 + w * 4 + e) / cpg) * 256 + (((unsigned)c[w] >> (8 * e)) & 0xff)], 1u);
            }
        }
        for (long i = beg + (nv << 4) + tid; i < end; i += 256) atomicAdd(&hg[((int)(i % C) / cpg) * 256 + x[i]], 1u);
    } else {
        for (long i = beg + tid; i < end; i += 256) atomicAdd(&hg[((int)(i % C) / cpg) * 256 + x[i]], 1u);
    }
    __syncthreads();
    for (int k = tid; k < G * 256; k += 256)
        if (hg[k]) atomicAdd(&hist[k], hg[k]);
}

__global__ __launch_bounds__(256) void q8_in_apply_nhwc_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, const uint8_t* __restrict__ lut_g, long n, int C,
                                                               int cpg, int G, int sh) {
    extern __shared__ uint8_t lg[];           // [G][256]
    const int tid = threadIdx.x;
    for (int k = tid; k < G * 256; k += 256) lg[k] = lut_g[k];
    __syncthreads();
    const long beg = (long)blockIdx.x * kInPiece, end = min(n, beg + kInPiece);
    if ((C & 15) == 0 && ((((uintptr_t)(x + beg)) | ((uintptr_t)(y + beg))) & 15) == 0) {
        const long nv = (end - beg) >> 4;
        for (long v = tid; v < nv; v += 256) {
            const long i0 = beg + (v << 4);
            const int c0 = (int)(i0 % C);
            const v4i c = *reinterpret_cast<const v4i*>(x + i0);
            v4i o;
            const int gb = sh >= 0 ? c0 >> sh : 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                unsigned r = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int g = sh >= 0 ? gb + ((w * 4 + e) >> sh) : (c0 + w * 4 + e) / cpg;
                    r |= (unsigned)lg[g * 256 + (((unsigned)c[w] >> (8 * e)) & 0xff)] << (8 * e);
                }
                o[w] = (int)r;
            }
            *reinterpret_cast<v4i*>(y + i0) = o;
        }
        for (long i = beg + (nv << 4) + tid; i < end; i += 256) y[i] = lg[((int)(i % C) / cpg) * 256 + x[i]];
    } else {
        for (long i = beg + tid; i < end; i += 256) y[i] = lg[((int)(i % C) / cpg) * 256 + x[i]];
    }
}

// ---- InstanceNormalization + affine chain as ONE table per channel ---------------------------------------------------------------------------------------
// x -> group table -> Mul(., g[c]) -> Add(., b[c]) [-> Sigmoid -> Mul] is a function of (channel, code): 256 values per channel.  One workgroup per group
// forms the group's normalisation table exactly as q8_in_lut_kernel does, then pushes every code through the affine chain of each of its channels with the
// very functions the separate launches use (q8_bin_op, the host-built sigmoid table) -> chan_lut [C][256]; the pass over the tensor is then a pure lookup
// (the chain costs ~40 integer instructions per code, the pass over a 512 x 512 x 128 tensor was VALU-bound at 0.5 TB/s).  Same codes by construction.
template <bool ACT>
__global__ __launch_bounds__(256) void q8_chan_lut_kernel(const unsigned* __restrict__ hist_g, uint8_t* __restrict__ chan_lut, long L, int n_scale,
                                                          const float* __restrict__ scale, const float* __restrict__ bias, float eps, float in_scale, int in_zp,
                                                          float n_out_scale, int n_out_zp, const uint8_t* __restrict__ g, const uint8_t* __restrict__ b,
                                                          const uint8_t* __restrict__ sig_lut, int cpg, Q8Bin qm, Q8Bin qa, Q8Bin qs) {
#pragma clang fp contract(off)
    __shared__ unsigned hist[256];
    __shared__ float deq[256];
    __shared__ double stat[2];
    const int row = blockIdx.x, c = threadIdx.x;
    hist[c] = hist_g[row * 256 + c];
    deq[c] = (float)(c - in_zp) * in_scale;
    __syncthreads();
    if (c == 0) {
        double mean = 0;
        for (int k = 0; k < 256; k++) mean += (double)hist[k] * (double)deq[k];
        mean /= (double)L;
        double var = 0;
        for (int k = 0; k < 256; k++) {
            const float dev = (float)((double)deq[k] - mean);
            var += (double)hist[k] * (double)(dev * dev);
        }
        var /= (double)L;
        stat[0] = mean;
        stat[1] = sqrt(var + (double)eps);
    }
    __syncthreads();
    const double sc = (double)scale[row % n_scale], bi = (double)bias[row % n_scale];
    const float v = (float)(sc * ((double)deq[c] - stat[0]) / stat[1] + bi);
    const float inv = 1.0f / n_out_scale;
    const int code = q8_quantize(v, inv, n_out_zp);
    for (int k = 0; k < cpg; k++) {
        const int ch = row * cpg + k;
        const unsigned v1 = q8_bin_op<1>(code, g[ch], qm);
        unsigned v2 = q8_bin_op<0>((int)v1, b[ch], qa);
        if (ACT) v2 = q8_bin_op<1>((int)v2, sig_lut[v2], qs);
        chan_lut[(long)ch * 256 + c] = (uint8_t)v2;
    }
}

// y[p][c] = chan_lut[c][x[p][c]] on NHWC codes.  A workgroup owns CB = 128 channels (their 32 KiB of tables in LDS) of a run of pixels; a lane looks up the
// 16 codes of one 16-byte vector.  C % 16 == 0, 16-byte aligned rows (the graphs' channel counts are multiples of 32).
constexpr int kChanBlock = 128;
__global__ __launch_bounds__(256) void q8_chan_apply_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, const uint8_t* __restrict__ chan_lut, long HW, int C,
                                                            int pix_per_wg) {
    __shared__ __attribute__((aligned(16))) uint8_t t[kChanBlock * 256];
    const int tid = threadIdx.x;
    const int cb0 = blockIdx.y * kChanBlock, cw = min(kChanBlock, C - cb0);      // this workgroup's channels
    {
        const v4i* __restrict__ src = reinterpret_cast<const v4i*>(chan_lut + (long)cb0 * 256);
        v4i* dst = reinterpret_cast<v4i*>(t);
        for (int k = tid; k < cw * 16; k += 256) dst[k] = src[k];
    }
    __syncthreads();
    const int vpp = cw >> 4;                                   // 16-byte vectors per pixel inside the channel block
    const long p0 = (long)blockIdx.x * pix_per_wg, p1 = min(HW, p0 + pix_per_wg);
    const int nv = (int)(p1 - p0) * vpp;
    for (int v = tid; v < nv; v += 256) {
        const int pv = v / vpp;
        const long p = p0 + pv;
        const int cl = (v - pv * vpp) << 4;
        const long off = p * C + cb0 + cl;
        const v4i c = *reinterpret_cast<const v4i*>(x + off);
        const uint8_t* __restrict__ tb = t + cl * 256;
        v4i o;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            unsigned r = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) r |= (unsigned)tb[(w * 4 + e) * 256 + (((unsigned)c[w] >> (8 * e)) & 0xff)] << (8 * e);
            o[w] = (int)r;
        }
        *reinterpret_cast<v4i*>(y + off) = o;
    }
}

// XNNPACK qu8 softmax over the last axis, one workgroup per row: t = host-built exp table (uint32[256]);
// y = min(255, ((t[x + 255 - max] << 8) + (sum >> 1)) / sum)
__global__ __launch_bounds__(256) void q8_softmax_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ y, long C, const unsigned* __restrict__ lut) {
    __shared__ unsigned t[256];
    __shared__ unsigned red[4];
    const uint8_t* __restrict__ xr = x + (long)blockIdx.x * C;
    uint8_t* __restrict__ yr = y + (long)blockIdx.x * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    t[tid] = lut[tid];
    unsigned mx = 0;
    for (long i = tid; i < C; i += 256) mx = max(mx, (unsigned)xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = max(max(red[0], red[1]), max(red[2], red[3]));
    __syncthreads();
    const unsigned off = 255u - mx;
    unsigned sum = 0;
    for (long i = tid; i < C; i += 256) sum += t[xr[i] + off];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += (unsigned)__shfl_xor((int)sum, o, 64);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    for (long i = tid; i < C; i += 256) {
        const unsigned long long q = (((unsigned long long)t[xr[i] + off] << 8) + (sum >> 1)) / sum;
        yr[i] = q > 255ull ? (uint8_t)255 : (uint8_t)q;
    }
}

inline unsigned grid_for(long work_items) {
    long blocks = (work_items + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;
    return (unsigned)blocks;
}

}  // namespace

extern "C" {

int osg_qu8_gemm(osg_ctx* ctx, const void* A, long lda, float a_scale, int a_zp, const void* B_nk, float b_scale, int b_zp, const float* bias_f32,
                 float out_scale, int out_zp, void* C, int M, int N, int K, int batch, long stride_a, long stride_b, long stride_c) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) OSG_FAIL(ctx, "osg_qu8_gemm: invalid shape of inputs");
    if (lda < K) OSG_FAIL(ctx, "osg_qu8_gemm: lda < K");
    if (batch > 65535) OSG_FAIL(ctx, "osg_qu8_gemm: batch too large");
    Q8Params p{};
    p.A = (const uint8_t*)A; p.Bt = (const uint8_t*)B_nk; p.C = (uint8_t*)C; p.bias = bias_f32;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.strideA = stride_a; p.strideB = stride_b; p.strideC = stride_c;
    p.a_zp = a_zp; p.b_zp = b_zp; p.out_zp = out_zp;
    p.ab_scale = a_scale * b_scale;
    p.out_scale = out_scale;
    return launch_q8(ctx, p, batch, false);
}

int osg_qu8_conv_tap_sums(osg_ctx* ctx, const void* w_ohwi, int Cout, int KH, int KW, int Cin, int* tap_sums) {
    if (Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0 || !w_ohwi || !tap_sums) OSG_FAIL(ctx, "osg_qu8_conv_tap_sums: invalid argument");
    hipLaunchKernelGGL(q8_tap_sums_kernel, dim3((unsigned)(Cout * KH * KW)), dim3(64), 0, ctx->compute, (const uint8_t*)w_ohwi, tap_sums, Cin);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_qu8_conv2d_nhwc_t(osg_ctx* ctx, const void* x, float x_scale, int x_zp, const void* w_ohwi, float w_scale, int w_zp, const float* bias_f32,
                          float out_scale, int out_zp, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl,
                          int pb, int pr, const int* tap_sums) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0) OSG_FAIL(ctx, "osg_qu8_conv2d_nhwc: invalid argument");
    const int Ho = (H + pt + pb - KH) / sh + 1, Wo = (W + pl + pr - KW) / sw + 1;
    if (Ho <= 0 || Wo <= 0) OSG_FAIL(ctx, "osg_qu8_conv2d_nhwc: empty output");
    Q8Params p{};
    p.A = (const uint8_t*)x; p.Bt = (const uint8_t*)w_ohwi; p.C = (uint8_t*)y; p.bias = bias_f32;
    p.M = N * Ho * Wo; p.N = Cout; p.K = KH * KW * Cin; p.lda = 0;
    p.a_zp = x_zp; p.b_zp = w_zp; p.out_zp = out_zp;
    p.ab_scale = x_scale * w_scale;
    p.out_scale = out_scale;
    p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.KW = KW; p.KH = KH; p.sh = sh; p.sw = sw; p.pt = pt; p.pl = pl;
    p.wtap = tap_sums;
    return launch_q8(ctx, p, 1, true);
}

int osg_qu8_conv2d_nhwc(osg_ctx* ctx, const void* x, float x_scale, int x_zp, const void* w_ohwi, float w_scale, int w_zp, const float* bias_f32,
                        float out_scale, int out_zp, void* y, int N, int H, int W, int Cin, int Cout, int KH, int KW, int sh, int sw, int pt, int pl,
                        int pb, int pr) {
    return osg_qu8_conv2d_nhwc_t(ctx, x, x_scale, x_zp, w_ohwi, w_scale, w_zp, bias_f32, out_scale, out_zp, y, N, H, W, Cin, Cout, KH, KW, sh, sw, pt, pl, pb, pr, nullptr);
}

int osg_qu8_lut(osg_ctx* ctx, const void* x, void* y, long n, const void* lut256) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(q8_lut_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->compute, (const uint8_t*)x, (uint8_t*)y, n, (const uint8_t*)lut256);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

// parameters of one uint8 Add / Mul exactly as osg_qu8_binary derives them (shared with the fused per-channel chain below)
static int make_q8bin(osg_ctx* ctx, bool mul, float a_scale, int a_zp, float b_scale, int b_zp, float out_scale, int out_zp, Q8Bin* out) {
    Q8Bin q{};
    q.a_zp = a_zp; q.b_zp = b_zp; q.out_zp = out_zp;
    if (mul) {
        q.scale = (a_scale * b_scale) / out_scale;
    } else {
        const float a_os = a_scale / out_scale, b_os = b_scale / out_scale;
        const float mx = a_os > b_os ? a_os : b_os;
        unsigned bits;
        memcpy(&bits, &mx, 4);
        const int exponent = (int)(bits >> 23) - 127;
        const int shift = 20 - exponent;
        if (shift < 1 || shift > 31) OSG_FAIL(ctx, "osg_qu8_binary: scale ratio out of the range of the fixed-point add");
        q.shift = (unsigned)shift;
        q.a_mult = (int)lrintf(ldexpf(a_os, shift));
        q.b_mult = (int)lrintf(ldexpf(b_os, shift));
        q.bias = (int)((1u << (shift - 1)) - (unsigned)(q.a_mult * a_zp) - (unsigned)(q.b_mult * b_zp));
    }
    *out = q;
    return 0;
}

// Mul(x, g[C]) -> Add(., b[C]) [-> Sigmoid -> Mul(., sigmoid)] on a tensor whose channel index is (i / inner) % C (NHWC: inner 1, NCHW: inner H*W):
// the GroupNorm affine (+ SiLU) of the uint8 graphs as ONE pass -- every stage's code is formed by the very functions the separate launches use
// (q8_bin_op, the host-built sigmoid table), in their order, so the output codes are those of the four launches.
int osg_qu8_affine_act(osg_ctx* ctx, const void* x, float x_scale, int x_zp, const void* g, float g_scale, int g_zp, float m_scale, int m_zp, const void* b,
                       float b_scale, int b_zp, float a_scale, int a_zp, const void* sig_lut, float s_scale, int s_zp, float o_scale, int o_zp, void* y, long n, int C,
                       long inner) {
    if (n <= 0) return 0;
    if (C <= 0 || inner <= 0) OSG_FAIL(ctx, "osg_qu8_affine_act: invalid shape");
    Q8Bin qm, qa, qs{};
    if (make_q8bin(ctx, true, x_scale, x_zp, g_scale, g_zp, m_scale, m_zp, &qm)) return 1;
    if (make_q8bin(ctx, false, m_scale, m_zp, b_scale, b_zp, a_scale, a_zp, &qa)) return 1;
    if (sig_lut && make_q8bin(ctx, true, a_scale, a_zp, s_scale, s_zp, o_scale, o_zp, &qs)) return 1;
    const dim3 grid(grid_for(n / 16 + 1)), block(256);
    if (sig_lut) hipLaunchKernelGGL((q8_affine_act_kernel<true, false>), grid, block, 0, ctx->compute, (const uint8_t*)x, (const uint8_t*)g, (const uint8_t*)b, (const uint8_t*)sig_lut, (uint8_t*)y, n, C, inner, qm, qa, qs, (const uint8_t*)nullptr, 1, 0);
    else hipLaunchKernelGGL((q8_affine_act_kernel<false, false>), grid, block, 0, ctx->compute, (const uint8_t*)x, (const uint8_t*)g, (const uint8_t*)b, (const uint8_t*)nullptr, (uint8_t*)y, n, C, inner, qm, qa, qs, (const uint8_t*)nullptr, 1, 0);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

// InstanceNormalization over NHWC (osg_qu8_instance_norm_nhwc) whose table lookup rides in the affine pass that follows it: histogram, tables, then ONE
// pass x -> table -> Mul -> Add [-> Sigmoid -> Mul].  n_* = the normalisation's own output parameters (the Mul's input).
int osg_qu8_norm_affine_act_nhwc(osg_ctx* ctx, const void* x, long HW, int C, int G, int n_scale, const float* scale, const float* bias, float eps, float x_scale,
                                 int x_zp, float n_out_scale, int n_out_zp, const void* g, float g_scale, int g_zp, float m_scale, int m_zp, const void* b, float b_scale,
                                 int b_zp, float a_scale, int a_zp, const void* sig_lut, float s_scale, int s_zp, float o_scale, int o_zp, void* y) {
    if (HW <= 0 || C <= 0 || G <= 0 || C % G || n_scale <= 0) OSG_FAIL(ctx, "osg_qu8_norm_affine_act_nhwc: invalid shape");
    if (G > 56) OSG_FAIL(ctx, "osg_qu8_norm_affine_act_nhwc: more than 56 groups do not fit the histogram in LDS");
    const size_t hist_bytes = (size_t)G * 256 * sizeof(unsigned), lut_bytes = (size_t)G * 256;
    if (osg_ensure_workspace(ctx, hist_bytes + (size_t)C * 256)) return 1;      // (group tables [G][256] or channel tables [C][256] behind the histogram)
    unsigned* hist = (unsigned*)ctx->ws;
    uint8_t* lut = (uint8_t*)ctx->ws + hist_bytes;
    OSG_HIP(ctx, hipMemsetAsync(hist, 0, hist_bytes, ctx->compute));
    const long n = HW * C, L = HW * (C / G);
    const unsigned pieces = (unsigned)((n + kInPiece - 1) / kInPiece);
    const int cpg = C / G;
    int sh = -1;
    if ((cpg & (cpg - 1)) == 0)
        for (sh = 0; (1 << sh) < cpg; sh++) {}
    hipLaunchKernelGGL(q8_in_hist_nhwc_kernel, dim3(pieces), dim3(256), hist_bytes, ctx->compute, (const uint8_t*)x, hist, n, C, cpg, G, sh);
    Q8Bin qm, qa, qs{};
    if (make_q8bin(ctx, true, n_out_scale, n_out_zp, g_scale, g_zp, m_scale, m_zp, &qm)) return 1;
    if (make_q8bin(ctx, false, m_scale, m_zp, b_scale, b_zp, a_scale, a_zp, &qa)) return 1;
    if (sig_lut && make_q8bin(ctx, true, a_scale, a_zp, s_scale, s_zp, o_scale, o_zp, &qs)) return 1;
    static const bool chain_per_code = getenv("OSG_QU8_NORM_CHAIN") != nullptr;     // A/B: the round-2 form (table per group, affine chain evaluated per code)
    if (!chain_per_code && (C & 15) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
        uint8_t* chan_lut = (uint8_t*)ctx->ws + hist_bytes;
        if (sig_lut) hipLaunchKernelGGL((q8_chan_lut_kernel<true>), dim3((unsigned)G), dim3(256), 0, ctx->compute, hist, chan_lut, L, n_scale, scale, bias, eps, x_scale, x_zp, n_out_scale, n_out_zp, (const uint8_t*)g, (const uint8_t*)b, (const uint8_t*)sig_lut, cpg, qm, qa, qs);
        else hipLaunchKernelGGL((q8_chan_lut_kernel<false>), dim3((unsigned)G), dim3(256), 0, ctx->compute, hist, chan_lut, L, n_scale, scale, bias, eps, x_scale, x_zp, n_out_scale, n_out_zp, (const uint8_t*)g, (const uint8_t*)b, (const uint8_t*)nullptr, cpg, qm, qa, qs);
        const int cblocks = (C + kChanBlock - 1) / kChanBlock;
        long ppw = HW * cblocks / 1024;                          // ~1024 workgroups when the tensor is big enough
        ppw = ppw < 64 ? 64 : ppw > 1024 ? 1024 : ppw;
        hipLaunchKernelGGL(q8_chan_apply_kernel, dim3((unsigned)((HW + ppw - 1) / ppw), (unsigned)cblocks), dim3(256), 0, ctx->compute, (const uint8_t*)x, (uint8_t*)y, (const uint8_t*)chan_lut, HW, C, (int)ppw);
        OSG_LAUNCH_CHECK(ctx);
        return 0;
    }
    hipLaunchKernelGGL(q8_in_lut_kernel, dim3((unsigned)G), dim3(256), 0, ctx->compute, hist, lut, L, n_scale, scale, bias, eps, x_scale, x_zp, n_out_scale, n_out_zp);
    const dim3 grid(grid_for(n / 16 + 1)), block(256);
    if (sig_lut) hipLaunchKernelGGL((q8_affine_act_kernel<true, true>), grid, block, lut_bytes, ctx->compute, (const uint8_t*)x, (const uint8_t*)g, (const uint8_t*)b, (const uint8_t*)sig_lut, (uint8_t*)y, n, C, 1L, qm, qa, qs, (const uint8_t*)lut, cpg, G);
    else hipLaunchKernelGGL((q8_affine_act_kernel<false, true>), grid, block, lut_bytes, ctx->compute, (const uint8_t*)x, (const uint8_t*)g, (const uint8_t*)b, (const uint8_t*)nullptr, (uint8_t*)y, n, C, 1L, qm, qa, qs, (const uint8_t*)lut, cpg, G);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_qu8_binary(osg_ctx* ctx, osg_binary_kind kind, const void* a, const long* a_shape, float a_scale, int a_zp, const void* b, const long* b_shape,
                   float b_scale, int b_zp, void* y, float out_scale, int out_zp, int rank) {
    if (rank < 1 || rank > kMaxRank) OSG_FAIL(ctx, "osg_qu8_binary: rank must be in [1,6]");
    if (kind != OSG_BIN_ADD && kind != OSG_BIN_MUL) OSG_FAIL(ctx, "osg_qu8_binary: only Add and Mul have a uint8 branch (reference :3977, :5105)");
    Q8Bcast p{};
    p.rank = rank;
    long as = 1, bs = 1, n = 1;
    for (int d = rank - 1; d >= 0; d--) {
        if (a_shape[d] != b_shape[d] && a_shape[d] != 1 && b_shape[d] != 1) OSG_FAIL(ctx, "osg_qu8_binary: shapes are not broadcastable");
        p.oshape[d] = a_shape[d] > b_shape[d] ? a_shape[d] : b_shape[d];
        p.astride[d] = a_shape[d] == 1 ? 0 : as;
        p.bstride[d] = b_shape[d] == 1 ? 0 : bs;
        as *= a_shape[d];
        bs *= b_shape[d];
        n *= p.oshape[d];
    }
    Q8Bin q{};
    q.a_zp = a_zp; q.b_zp = b_zp; q.out_zp = out_zp;
    // fast shapes: `full` operand x (same shape | operand repeating with a period)
    long an = 1, bn = 1;
    for (int d = 0; d < rank; d++) { an *= a_shape[d]; bn *= b_shape[d]; }
    int mode = -1;             // 0: same shape, 1: b periodic, 2: a periodic (operands swapped below)
    long inner = 1, period = 1;
    auto periodic = [&](const long* sh, long cnt) {     // the non-1 dims of `sh` form one contiguous run matching the output there
        int lo = -1, hi = -1;
        for (int d = 0; d < rank; d++)
            if (sh[d] != 1) { if (lo < 0) lo = d; hi = d; }
        if (lo < 0) { inner = 1; period = 1; return true; }
        long per = 1;
        for (int d = lo; d <= hi; d++) { if (sh[d] != p.oshape[d]) return false; per *= sh[d]; }
        long in = 1;
        for (int d = hi + 1; d < rank; d++) in *= p.oshape[d];
        inner = in; period = per;
        return per == cnt;
    };
    const bool al = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) == 0);
    if (al && an == n && bn == n) mode = 0;
    else if (al && an == n && periodic(b_shape, bn)) mode = 1;
    else if (al && bn == n && periodic(a_shape, an)) mode = 2;
    const uint8_t *pa = (const uint8_t*)a, *pb = (const uint8_t*)b;
    if (mode == 2) {           // Add and Mul commute: swap the operands together with their parameters
        std::swap(pa, pb);
        std::swap(a_scale, b_scale);
        std::swap(a_zp, b_zp);
        q.a_zp = a_zp; q.b_zp = b_zp;
        mode = 1;
    }
    if (kind == OSG_BIN_MUL) {
        q.scale = (a_scale * b_scale) / out_scale;
        if (mode == 0) hipLaunchKernelGGL((q8_binary_fast_kernel<1, 0>), dim3(grid_for(n / 16 + 1)), dim3(256), 0, ctx->compute, pa, pb, (uint8_t*)y, n, inner, period, q);
        else if (mode == 1) hipLaunchKernelGGL((q8_binary_fast_kernel<1, 1>), dim3(grid_for(n / 16 + 1)), dim3(256), 0, ctx->compute, pa, pb, (uint8_t*)y, n, inner, period, q);
        else hipLaunchKernelGGL(q8_binary_kernel<1>, dim3(grid_for(n)), dim3(256), 0, ctx->compute, (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)y, n, p, q);
    } else {
        // XNNPACK qu8 add: the two input/output scale ratios as integer multipliers, 20 bits for the larger one, rounding folded into the bias
        const float a_os = a_scale / out_scale, b_os = b_scale / out_scale;
        const float mx = a_os > b_os ? a_os : b_os;
        unsigned bits;
        memcpy(&bits, &mx, 4);
        const int exponent = (int)(bits >> 23) - 127;
        const int shift = 20 - exponent;
        if (shift < 1 || shift > 31) OSG_FAIL(ctx, "osg_qu8_binary: scale ratio out of the range of the fixed-point add");
        q.shift = (unsigned)shift;
        q.a_mult = (int)lrintf(ldexpf(a_os, shift));
        q.b_mult = (int)lrintf(ldexpf(b_os, shift));
        q.bias = (int)((1u << (shift - 1)) - (unsigned)(q.a_mult * a_zp) - (unsigned)(q.b_mult * b_zp));
        if (mode == 0) hipLaunchKernelGGL((q8_binary_fast_kernel<0, 0>), dim3(grid_for(n / 16 + 1)), dim3(256), 0, ctx->compute, pa, pb, (uint8_t*)y, n, inner, period, q);
        else if (mode == 1) hipLaunchKernelGGL((q8_binary_fast_kernel<0, 1>), dim3(grid_for(n / 16 + 1)), dim3(256), 0, ctx->compute, pa, pb, (uint8_t*)y, n, inner, period, q);
        else hipLaunchKernelGGL(q8_binary_kernel<0>, dim3(grid_for(n)), dim3(256), 0, ctx->compute, (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)y, n, p, q);
    }
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_qu8_instance_norm(osg_ctx* ctx, const void* x, void* y, int rows, long L, int n_scale, const float* scale, const float* bias, float eps,
                          float in_scale, int in_zp, float out_scale, int out_zp) {
    if (rows <= 0 || L <= 0 || n_scale <= 0) OSG_FAIL(ctx, "osg_qu8_instance_norm: invalid shape");
    if (rows > 65535) OSG_FAIL(ctx, "osg_qu8_instance_norm: too many rows");
    const size_t hist_bytes = (size_t)rows * 256 * sizeof(unsigned), lut_bytes = (size_t)rows * 256;
    if (osg_ensure_workspace(ctx, hist_bytes + lut_bytes)) return 1;
    unsigned* hist = (unsigned*)ctx->ws;
    uint8_t* lut = (uint8_t*)ctx->ws + hist_bytes;
    OSG_HIP(ctx, hipMemsetAsync(hist, 0, hist_bytes, ctx->compute));
    const unsigned pieces = (unsigned)((L + kInPiece - 1) / kInPiece);
    hipLaunchKernelGGL(q8_in_hist_kernel, dim3(pieces, (unsigned)rows), dim3(256), 0, ctx->compute, (const uint8_t*)x, hist, L);
    hipLaunchKernelGGL(q8_in_lut_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->compute, hist, lut, L, n_scale, scale, bias, eps, in_scale, in_zp, out_scale, out_zp);
    hipLaunchKernelGGL(q8_in_apply_kernel, dim3(pieces, (unsigned)rows), dim3(256), 0, ctx->compute, (const uint8_t*)x, (uint8_t*)y, lut, L);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_qu8_instance_norm_nhwc(osg_ctx* ctx, const void* x, void* y, long HW, int C, int G, int n_scale, const float* scale, const float* bias, float eps,
                               float in_scale, int in_zp, float out_scale, int out_zp) {
    if (HW <= 0 || C <= 0 || G <= 0 || C % G || n_scale <= 0) OSG_FAIL(ctx, "osg_qu8_instance_norm_nhwc: invalid shape");
    if (G > 56) OSG_FAIL(ctx, "osg_qu8_instance_norm_nhwc: more than 56 groups do not fit the histogram in LDS");
    const size_t hist_bytes = (size_t)G * 256 * sizeof(unsigned), lut_bytes = (size_t)G * 256;
    if (osg_ensure_workspace(ctx, hist_bytes + lut_bytes)) return 1;
    unsigned* hist = (unsigned*)ctx->ws;
    uint8_t* lut = (uint8_t*)ctx->ws + hist_bytes;
    OSG_HIP(ctx, hipMemsetAsync(hist, 0, hist_bytes, ctx->compute));
    const long n = HW * C, L = HW * (C / G);
    const unsigned pieces = (unsigned)((n + kInPiece - 1) / kInPiece);
    const int cpg = C / G;
    int sh = -1;
    if ((cpg & (cpg - 1)) == 0)
        for (sh = 0; (1 << sh) < cpg; sh++) {}
    hipLaunchKernelGGL(q8_in_hist_nhwc_kernel, dim3(pieces), dim3(256), hist_bytes, ctx->compute, (const uint8_t*)x, hist, n, C, C / G, G, sh);
    hipLaunchKernelGGL(q8_in_lut_kernel, dim3((unsigned)G), dim3(256), 0, ctx->compute, hist, lut, L, n_scale, scale, bias, eps, in_scale, in_zp, out_scale, out_zp);
    hipLaunchKernelGGL(q8_in_apply_nhwc_kernel, dim3(pieces), dim3(256), lut_bytes, ctx->compute, (const uint8_t*)x, (uint8_t*)y, lut, n, C, C / G, G, sh);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_qu8_softmax_last(osg_ctx* ctx, const void* x, void* y, long rows, long C, const void* lut_u32_256) {
    if (rows <= 0 || C <= 0) return 0;
    if (rows > 2147483647L) OSG_FAIL(ctx, "osg_qu8_softmax_last: too many rows");
    hipLaunchKernelGGL(q8_softmax_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->compute, (const uint8_t*)x, (uint8_t*)y, C, (const unsigned*)lut_u32_256);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
