"""Generate tests/golden/sd_loop.npz: the reference APPLICATION's own denoising loop (src/sd.cpp diffusion_solver + CFGDenoiser_CompVisDenoiser +
src/samplers.h Euler-Ancestral, compiled as they lie into the oracle: oracle/ref_sd.cpp) run for 3 steps, CFG 7, on a narrow synthetic UNet
with the SD 1.5 interface the app hard-codes (sample [1,4,64,64], context [1,77,768]).  Stored: contexts, the reference's initial latent
and per-step ancestral noise (randn_4_w_h with the app's srand/rand seed walk), the final latents of 1 and of 2 batched images."""
import ctypes
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.synth import sd_unet  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402
from oracle import ref as oref  # noqa: E402

IFACE = sd_unet.UNetConfig(block_out=(32, 64), transformer_depth=(1, 1), heads=2, groups=8, latent=64, name="sd15iface")
SEED, STEPS = 9, 3


def build_micro_unet(sink):
    """the cheapest graph with the app's UNet interface (timestep [1], sample [1,4,64,64], encoder_hidden_states [1,77,768] -> [1,4,64,64]):
    conv3x3(sample) + timestep * a[4,1,1] + mean_tokens(ctx W)[1,4,1,1] -- milliseconds per pass, so that the reference application's
    20-step loop (schedule, sigma_to_t, CFG, ancestral update, noise walk) can be compared end to end in a CPU test"""
    from onnxstream_amd.synth.graph import GraphBuilder
    g = GraphBuilder(sink, seed=77)
    t = g.input("timestep", (1,))
    x = g.input("sample", (1, 4, 64, 64))
    c = g.input("encoder_hidden_states", (1, 77, 768))
    y = g.conv("/conv", x, 4, 3, std=0.15)
    a = g.weight("/t.scale", g.randn((4, 1, 1), 1e-4), allow_quant=False)
    y = g.binary("/add_t", "Add", y, g.binary("/mul_t", "Mul", t, a))
    m = g.matmul_w("/ctx/MatMul", c, 4)
    m = g.transpose("/ctx/T", m, (0, 2, 1))
    m = g.op("/ctx/ReduceMean", "ReduceMean", [m], (1, 4, 1), {"axes": "-1", "keepdims": "1"})
    m = g.reshape("/ctx/Reshape", m, (1, 4, 1, 1))
    g.op("/out", "Add", [y, m], (1, 4, 64, 64), out_names=["out_sample"])
    g.finish()


def ref_lib():
    lib = ctypes.CDLL(oref.REF_LIB)
    lib.ref_sd_diffusion_solver.restype = ctypes.c_char_p
    lib.ref_sd_diffusion_solver.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint] + [ctypes.c_void_p] * 3
    lib.ref_sd_set_sampler.argtypes = [ctypes.c_int]      # 0 = euler_a (the app's default), 1 = euler
    lib.ref_sd_set_sampler.restype = None
    lib.ref_sd_randn.argtypes = [ctypes.c_int, ctypes.c_void_p]
    lib.ref_sd_step_noise_seed.argtypes = [ctypes.c_int]
    lib.ref_sd_step_noise_seed.restype = ctypes.c_int
    return lib


def contexts():
    rng = np.random.default_rng(5)
    return rng.standard_normal((77, 768), dtype=np.float32), rng.standard_normal((77, 768), dtype=np.float32)


def ref_randn(lib, seed):
    out = np.empty((1, 4, 64, 64), np.float32)
    lib.ref_sd_randn(seed, out.ctypes.data)
    return out


def ref_noise_walk(lib, seed, steps):
    """initial latent + ancestral noises as diffusion_solver / process_sample draw them for an image started with `seed`"""
    init = ref_randn(lib, seed % 1000)
    return init, [ref_randn(lib, lib.ref_sd_step_noise_seed(seed + i)) for i in range(steps)]


def ref_loop(lib, models_dir, num, threads=0, steps=STEPS):
    cond, uncond = contexts()
    out = np.zeros((num, 4, 64, 64), np.float32)
    err = lib.ref_sd_diffusion_solver(models_dir.encode(), SEED, steps, num, threads or oref.usable_cores(), cond.ctypes.data, uncond.ctypes.data, out.ctypes.data)
    if err:
        raise RuntimeError(err.decode())
    return out


if __name__ == "__main__":
    assert oref.available()
    lib = ref_lib()
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d + "unet_fp16/"), IFACE)
        lat1 = ref_loop(lib, d, 1, threads=1)
        lat2 = ref_loop(lib, d, 2, threads=1)
        build_micro_unet(DirSink(d + "micro/unet_fp16/"))
        lat20 = ref_loop(lib, d + "micro/", 1, threads=1, steps=20)
    assert np.array_equal(lat1[0], lat2[0])
    cond, uncond = contexts()
    init0, noise0 = ref_noise_walk(lib, SEED, STEPS)
    init1, noise1 = ref_noise_walk(lib, SEED + 1, STEPS)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "sd_loop.npz"), latents=lat2, init=np.concatenate([init0, init1]),
                        noise=np.stack([np.concatenate([a, b]) for a, b in zip(noise0, noise1)]), seed=np.asarray(SEED), steps=np.asarray(STEPS), latents20_micro=lat20)
    print("sd_loop", lat2.shape, float(np.abs(lat2).max()), float(lat2.std()))
