// libosgpu: the per-step host arithmetic of the reference's denoising loop, moved next to the UNet so that a whole 20-step image can be
// enqueued without one host round trip:
//   osg_sampler_prepare     <- CFGDenoiser_CompVisDenoiser's input scaling  x * c_in  and the timestep broadcast (reference src/sd.cpp:1427-1470)
//   osg_sampler_cfg_euler_a <- eps -> denoised (x + eps * c_out), the CFG combine (uncond + g * (cond - uncond), src/sd.cpp:1545-1556) and
//                              the Euler-Ancestral update (src/samplers.h:1430-1449: the branch the shipped `#define ORIGINAL_SAMPLER_ALGORITHMS 1`,
//                              samplers.h:66, selects:  x = x + ((x - d) / sigma_i) * (sigma_down - sigma_i) + r * sigma_up)
// fp32 throughout, in the reference's operation order with every multiply and add rounded separately (no fma contraction), so the device
// loop reproduces the host loop bit for bit (tests/test_pipeline.py).
#include "osg_common.h"

// HIP compiles with -ffp-contract=fast (and its __fmul_rn/__fadd_rn are plain operators defined in a header): without the pragma inside
// the kernels a*b+c becomes one fma (one rounding) and the loop drifts from the host arithmetic by an ulp per step.

namespace {

__global__ __launch_bounds__(256) void sampler_prepare_kernel(const float* __restrict__ x, float* __restrict__ sample, float* __restrict__ timestep,
                                                              int prompts, long L, float c_in, float t, long t_per_sample) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)prompts * L;
    if (i < total) {
        const long p = i / L, e = i - p * L;
        const float v = x[i] * c_in;
        sample[(2 * p) * L + e] = v;          // pushes 2p (cond) and 2p+1 (uncond) see the same scaled latent
        sample[(2 * p + 1) * L + e] = v;
    }
    if (i < 2L * prompts * t_per_sample) timestep[i] = t;
}

__global__ __launch_bounds__(256) void sampler_cfg_euler_a_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                                                                  int prompts, long L, float c_out, float guidance, float sigma, float d_sigma, float sigma_up, float clip) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)prompts * L) return;
    const long p = i / L, e = i - p * L;
    const float xv = x[i];
    const float pc = eps[(2 * p) * L + e] * c_out, pu = eps[(2 * p + 1) * L + e] * c_out;
    const float den_c = pc + xv;
    const float den_u = pu + xv;
    const float gd = guidance * (den_c - den_u);
    const float den = den_u + gd;
    const float dd = (xv - den) / sigma;          // IEEE division (hipcc keeps fp32 divides correctly rounded), one rounding per operation
    const float st = dd * d_sigma;
    float nx = xv + st;
    if (noise) {
        const float nz = noise[i] * sigma_up;
        nx = nx + nz;
    }
    if (clip > 0.f) nx = fminf(fmaxf(nx, -clip), clip);
    x[i] = nx;
}

}  // namespace

extern "C" {

int osg_sampler_prepare(osg_ctx* ctx, const float* x, float* sample, float* timestep, int prompts, long L, float c_in, float t, long t_per_sample) {
    if (prompts <= 0 || L <= 0) return 0;
    const long total = (long)prompts * L;
    const long n = total > 2L * prompts * t_per_sample ? total : 2L * prompts * t_per_sample;
    hipLaunchKernelGGL(sampler_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->compute, x, sample, timestep, prompts, L, c_in, t,
                       t_per_sample);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

int osg_sampler_cfg_euler_a(osg_ctx* ctx, float* x, const float* eps, const float* noise, int prompts, long L, float c_out, float guidance,
                            float sigma, float d_sigma, float sigma_up, float clip) {
    if (prompts <= 0 || L <= 0) return 0;
    const long total = (long)prompts * L;
    hipLaunchKernelGGL(sampler_cfg_euler_a_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->compute, x, eps, noise, prompts, L, c_out,
                       guidance, sigma, d_sigma, sigma_up, clip);
    OSG_LAUNCH_CHECK(ctx);
    return 0;
}

}  // extern "C"
