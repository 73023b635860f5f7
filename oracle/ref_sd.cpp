// TEST INFRASTRUCTURE ONLY (oracle).  Never linked into the product libraries.
//
// ref_sd.cpp -- the reference's own txt2img application code as a callable: src/sd.cpp is #included WHERE IT LIES (its main() renamed,
// nothing copied), so that `diffusion_solver` (src/sd.cpp:1574-1780: sigma schedule :1597-1609, initial latent :1595/:1611-1612) with
// `CFGDenoiser_CompVisDenoiser` (:1397-1559: c_in / c_out / sigma_to_t :1399-1434, the CFG-7 combine :1545-1556) and the Euler-Ancestral
// update of src/samplers.h (the branch the shipped `#define ORIGINAL_SAMPLER_ALGORITHMS 1` selects, :1431-1449) run as COMPILED REFERENCE
// CODE over a UNet directory of ours.  This is what pins onnxstream_amd/pipeline.py and osg_sampler.hip: no arithmetic of the sampler
// or of the CFG combine is restated on the oracle side.
//
// sd.cpp's SD 1.5 branch hard-codes the tensor shapes it pushes (sample {1,4,64,64}, context {1,77,768}: :1468, :1474), so the UNet
// behind `<models>/unet_fp16/model.txt` must have that INTERFACE (its width/depth are free: the test uses a narrow synthetic one).
//
// cpuinfo: sd.cpp includes cpuinfo.h only to print whether fp16 is "detected" (:186-203); torch ships the header but does not export
// the library's symbols, so the three calls are answered here.
#define CPUINFO_H
static inline bool cpuinfo_initialize() { return true; }
static inline bool cpuinfo_has_x86_avx2() { return true; }
static inline bool cpuinfo_has_arm_neon_fp16_arith() { return false; }

#define USE_ONNXSTREAM 1
#define main onnxstream_reference_sd_main
#include "sd.cpp"
#undef main

static sampler_type g_ref_sampler = EULER_A;

extern "C" {

// 0 = "euler_a" (the app's default), 1 = "euler": the sampler diffusion_solver dispatches to (src/sd.cpp:1687, --sampler)
void ref_sd_set_sampler(int euler) { g_ref_sampler = euler ? EULER : EULER_A; }

// Runs `steps` denoising steps of the reference app (default sampler Euler-A, CFG 7) for `num` images batched like `--num` and writes the
// final latents [num,4,64,64].  cond / uncond: [77,768] fp32.  Returns NULL or the exception text.
const char* ref_sd_diffusion_solver(const char* models_path_with_slash, int seed, int steps, int num, unsigned threads, const float* cond,
                                    const float* uncond, float* latents_out) {
    static thread_local std::string err;
    try {
        g_main_args.m_path_with_slash = models_path_with_slash;
        g_main_args.m_latw = g_main_args.m_lath = 64;
        g_main_args.m_num = std::to_string(num);
        g_main_args.m_sampler = g_ref_sampler;
        n_threads = threads;
        ncnn::Mat c(768, 77, 1, (void*)cond), uc(768, 77, 1, (void*)uncond);
        std::vector<ncnn::Mat> samples;
        {
            SDCoroState coro_state;
            samples = coro_state.run<ncnn::Mat>([&]() { return diffusion_solver(seed + (int)coro_state.batch_index, steps, c, uc, std::string(), nullptr, coro_state); });
        }
        for (size_t i = 0; i < samples.size(); i++) memcpy(latents_out + i * 4 * 64 * 64, (float*)samples[i], 4 * 64 * 64 * sizeof(float));
        return nullptr;
    } catch (const std::exception& e) {
        err = e.what();
        return err.c_str();
    }
}

// one call of CFGDenoiser_CompVisDenoiser (src/sd.cpp:1397-1559) on a given latent: x [4,64,64] -> denoised [4,64,64] (CFG 7 of cond / uncond)
const char* ref_sd_cfg_denoise(const char* models_path_with_slash, unsigned threads, const float* x, float sigma, const float* cond, const float* uncond,
                               float* denoised_out) {
    static thread_local std::string err;
    try {
        g_main_args.m_path_with_slash = models_path_with_slash;
        g_main_args.m_latw = g_main_args.m_lath = 64;
        g_main_args.m_num = "1";
        n_threads = threads;
        ncnn::Mat c(768, 77, 1, (void*)cond), uc(768, 77, 1, (void*)uncond), xm(64, 64, 4, (void*)x);
        SDCoroState coro_state;
        coro_state.model.m_use_fp16_arithmetic = true;
        coro_state.model.m_fuse_ops_in_attention = true;
        coro_state.model.read_file((g_main_args.m_path_with_slash + "unet_fp16/model.txt").c_str());
        static float table[1000];
        {   // the app's literal table lives inside diffusion_solver; CFGDenoiser only needs it for sigma_to_t, which the caller fixes by passing log(sigma) = table[999]
            for (int i = 0; i < 1000; i++) table[i] = -1e9f;
            table[998] = std::log(sigma) - 1.0f;
            table[999] = std::log(sigma);
        }
        ncnn::Net net;
        auto res = coro_state.run<ncnn::Mat>([&]() { return CFGDenoiser_CompVisDenoiser(net, table, xm, sigma, c, uc, nullptr, coro_state); });
        memcpy(denoised_out, (float*)res[0], 4 * 64 * 64 * sizeof(float));
        return nullptr;
    } catch (const std::exception& e) {
        err = e.what();
        return err.c_str();
    }
}

// the noise the reference draws: randn_4_w_h(seed, 64, 64) (src/sd.cpp:1366: mt19937 + normal_distribution<float>) -> [4,64,64]
void ref_sd_randn(int seed, float* out) {
    ncnn::Mat m = randn_4_w_h(seed, 64, 64);
    memcpy(out, (float*)m, 4 * 64 * 64 * sizeof(float));
}

// the rand() stream process_sample consumes for the ancestral noise: std::srand(seed++); rand() % 1000 per step (src/samplers.h:1436-1437)
int ref_sd_step_noise_seed(int seed_at_step) {
    std::srand(seed_at_step);
    return rand() % 1000;
}

}  // extern "C"
