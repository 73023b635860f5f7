"""txt2img harness: the denoising loop + VAE decode of the reference app, host side (reference src/sd.cpp ``diffusion_solver``
:1574-1780, ``CFGDenoiser_CompVisDenoiser`` :1397-1559, Euler-Ancestral ``src/samplers.h`` :1430-1472, ``decoder_solver`` :1174-1256).

It drives ANY library that exports the reference's model_* C API through ``bindings.Model`` -- the HIP backend
(``libonnxstream_amd.so``) or the reference oracle -- which is what makes it a parity harness: same schedule, same CFG
combine (scale 7, hard-coded in the reference), same sampler arithmetic, only the UNet/VAE executor differs.
With the HIP backend the cond and uncond samples are pushed under the same names and run as ONE batch-2 pass
(the reference's ``m_batch``); with the reference library they run back to back, as ``sd.cpp`` does without ``--num``.

Out of scope (host glue of the app): tokenizer, text encoder, PNG writer.  The text context is an input.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np

from .bindings import Model

f32 = np.float32


def _libm_float_fn(name):
    """The reference app computes its schedule with the C library's FLOAT functions (std::exp / std::log on float: expf / logf, src/sd.cpp:1402,
    :1608).  numpy's float32 exp / log are its own SIMD kernels and differ from glibc's in the last bit for a third of the table
    (sigma[0] = 14.614644 vs 14.614643) -- one ulp of sigma flips f16 roundings of the UNet input and moves a whole pass by 1e-3.  So the
    harness calls the same libm; without one, the correctly rounded value (double function, rounded once) stands in."""
    import ctypes
    import ctypes.util
    import math
    try:
        fn = getattr(ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6"), name + "f")
        fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
        return lambda v: f32(fn(float(v)))
    except (OSError, AttributeError):
        dbl = getattr(math, name)
        return lambda v: f32(dbl(float(v)))


expf, logf = _libm_float_fn("exp"), _libm_float_fn("log")


def log_sigmas_table() -> np.ndarray:
    """The 1000-entry table hard-coded at src/sd.cpp:1591: log(sqrt((1-acp)/acp)) of SD's scaled-linear beta schedule
    (beta 0.00085 -> 0.012 over 1000 steps), recomputed the way it was made -- alphas_cumprod in float64, stored as float32, sigma and log
    in float32.  The reference's literals are the float32 results of whatever libm produced them: this recomputation hits 827 of the
    1000 entries exactly and is 1 float32 ulp (<= 2.4e-7) off in the others (tests/test_pipeline.py checks that against the source)."""
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    acp = np.cumprod(1.0 - betas).astype(f32)
    return np.log(np.sqrt((f32(1) - acp) / acp)).astype(f32)


def sigma_schedule(steps: int, log_sigmas: np.ndarray) -> np.ndarray:
    """t_to_sigma over linspace(999, 0, steps) + a trailing 0 (src/sd.cpp:1597-1612)."""
    sig = np.empty(steps + 1, f32)
    delta = f32(-999.0) / f32(steps - 1) if steps > 1 else f32(0)      # float delta = -999.0f / (step - 1)
    for i in range(steps):
        t = f32(999.0 + float(f32(f32(i) * delta)))                   # float t = 999.0 + i * delta  (float product, double sum, one rounding)
        lo, hi = int(np.floor(t)), int(np.ceil(t))
        w = f32(t - lo)
        sig[i] = expf(f32(f32(f32(1) - w) * log_sigmas[lo]) + f32(w * log_sigmas[hi]))
    sig[steps] = 0.0
    return sig


def sigma_to_t(sigma: float, log_sigmas: np.ndarray) -> float:
    """src/sd.cpp:1403-1425."""
    ls = logf(f32(sigma))
    dists = np.cumsum((ls - log_sigmas >= 0).astype(np.int64))
    low = min(int(np.argmax(dists)), 1000 - 2)
    high = low + 1
    lo, hi = log_sigmas[low], log_sigmas[high]
    w = f32(f32(lo - ls) / f32(lo - hi))                      # float arithmetic throughout, as :1420-1422
    w = max(f32(0), min(f32(1), w))
    return float(f32(f32(f32(1) - w) * f32(low)) + f32(w * f32(high)))


def save_latents(path: str, latents: np.ndarray) -> None:
    """`sd --save-latents FILE` (src/sd.cpp:2325-2327): the first sample's latents as raw little-endian float32, [4, h, w]."""
    np.ascontiguousarray(np.asarray(latents, f32).reshape((-1,) + tuple(np.asarray(latents).shape[-3:]))[0], "<f4").tofile(path)


def load_latents(path: str, h: int = 64, w: int = 64) -> np.ndarray:
    """`sd --decode-latents FILE` (src/sd.cpp:3212-3245): a raw float32 file back as [1, 4, h, w] (the app checks the size against 4*h*w too)."""
    v = np.fromfile(path, "<f4")
    if v.size != 4 * h * w:
        raise ValueError("Invalid size of latents file.")
    return v.reshape(1, 4, h, w).astype(f32)


class Txt2Img:
    def __init__(self, library: str, unet_dir: str, vae_dir: Optional[str], batched: bool = True, device: int = 0,
                 names: Dict[str, str] = None, fusion: Optional[int] = None, threads: int = 0, autotune: Optional[bool] = None):
        self.batched = batched
        self.names = dict(timestep="timestep", sample="sample", ctx="encoder_hidden_states", out="out_sample", vae_in="input.1",
                          vae_out="out_image")
        if names:
            self.names.update(names)
        self.log_sigmas = log_sigmas_table()
        self._t_cache: Dict[float, float] = {}
        self.unet = Model(library, threads, "ram+nocache")
        self.vae = Model(library, threads, "ram+nocache") if vae_dir else None
        for m, d in ((self.unet, unet_dir), (self.vae, vae_dir)):
            if m is None:
                continue
            if batched:
                m._set_option("hip_device", device)
                if fusion is not None:
                    m._set_option("hip_fusion_level", fusion)
                if autotune is not None:
                    m._set_option("hip_autotune", int(bool(autotune)))
            m.read_file(d + "model.txt")
        self._configured: Dict[int, bool] = {}
        self._dev_ready: Dict[tuple, bool] = {}
        self.last_loop_ms = 0.0

    def close(self):
        self.unet.close()
        if self.vae:
            self.vae.close()

    def _run(self, m: Model, pushes: List[Dict[str, np.ndarray]], out: str) -> List[np.ndarray]:
        if self.batched:
            for ins in pushes:
                for k, v in ins.items():
                    m.add_tensor(k, v if (v.dtype == f32 and v.flags.c_contiguous) else np.ascontiguousarray(v, f32))
            if not self._configured.get(id(m)):
                m.set_use_fp16_arithmetic(True)
                m.set_fuse_ops_in_attention(True)
                self._configured[id(m)] = True
            m.run()
            res = [m.get_tensor(out, i)[0] for i in range(len(pushes))]
            m.clear_tensors()
            return res
        res = []
        for ins in pushes:            # the reference C API takes fp32 inputs only while fp16 arithmetic is off (see oracle/ref.py)
            m.set_use_fp16_arithmetic(False)
            for k, v in ins.items():
                m.add_tensor(k, np.ascontiguousarray(v, f32))
            m.set_use_fp16_arithmetic(True)
            m.set_fuse_ops_in_attention(True)
            m.run()
            res.append(m.get_tensor(out)[0])
            m.clear_tensors()
        return res

    def denoise(self, x: np.ndarray, sigma: float, cond: np.ndarray, uncond: np.ndarray, guidance: float = 7.0,
                extra_cond: Optional[Dict[str, np.ndarray]] = None, extra_uncond: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
        """CFGDenoiser_CompVisDenoiser: eps-prediction wrapped as a denoiser, then the CFG combine (src/sd.cpp:1397-1559)."""
        n = self.names
        c_out = f32(-1.0 * sigma)
        c_in = f32(1.0 / np.sqrt(f32(sigma) * f32(sigma) + 1))
        if sigma not in self._t_cache:            # the schedule revisits the same 20 sigmas for every image
            self._t_cache[sigma] = sigma_to_t(sigma, self.log_sigmas)
        t = f32(self._t_cache[sigma])
        xin = (x * c_in).astype(f32)
        if xin.shape[0] > 1 and isinstance(cond, (list, tuple)):
            # several prompts at the same step of the schedule (the reference's `--num N` batching, src/sd.cpp:1098-1161): 2N samples
            # pushed under the same names -> ONE batched pass; prompt p = pushes 2p (cond) and 2p+1 (uncond)
            pushes = []
            for p_i in range(xin.shape[0]):
                for c in (cond[p_i], uncond[p_i]):
                    pushes.append({n["timestep"]: np.asarray([t], f32), n["sample"]: xin[p_i:p_i + 1], n["ctx"]: c})
            eps = self._run(self.unet, pushes, n["out"])
            den_c = np.concatenate(eps[0::2]) * c_out + x
            den_u = np.concatenate(eps[1::2]) * c_out + x
            return (den_u + f32(guidance) * (den_c - den_u)).astype(f32)
        pushes = [{n["timestep"]: np.asarray([t], f32), n["sample"]: xin, n["ctx"]: c} for c in (cond, uncond)]
        # SDXL micro-conditioning (text_embeds [1,1280], time_ids [1,6]; reference src/sd.cpp:1488-1516) rides along per branch
        for push, extra in zip(pushes, (extra_cond, extra_uncond)):
            if extra:
                push.update(extra)
        eps_c, eps_u = self._run(self.unet, pushes, n["out"])
        den_c = eps_c * c_out + x
        den_u = eps_u * c_out + x
        return (den_u + f32(guidance) * (den_c - den_u)).astype(f32)

    @staticmethod
    def ancestral_step_scalars(s_i, s_n):
        """(sigma_up, sigma_down) of one Euler-Ancestral step exactly as the reference's ACTIVE branch computes them -- samplers.h:66
        ships `#define ORIGINAL_SAMPLER_ALGORITHMS 1`, i.e. src/samplers.h:1431-1433, all in float:
            sigma_up   = min(s1, sqrt(s1 * s1 * (s0 * s0 - s1 * s1) / (s0 * s0)));   sigma_down = sqrt(s1 * s1 - sigma_up * sigma_up)"""
        s0, s1 = f32(s_i), f32(s_n)
        with np.errstate(invalid="ignore"):
            sigma_up = min(s1, np.sqrt(f32(f32(f32(s1 * s1) * f32(f32(s0 * s0) - f32(s1 * s1))) / f32(s0 * s0))))
            sigma_down = np.sqrt(f32(f32(s1 * s1) - f32(sigma_up * sigma_up)))
        return f32(sigma_up), f32(sigma_down)

    def sample(self, cond: np.ndarray, uncond: np.ndarray, steps: int = 20, seed: int = 42, latent_shape=(1, 4, 64, 64),
               on_step: Optional[Callable[[int, np.ndarray], None]] = None, init_latent: Optional[np.ndarray] = None,
               step_noise: Optional[Callable[[int], np.ndarray]] = None, sampler: str = "euler_a") -> np.ndarray:
        """diffusion_solver (src/sd.cpp:1574-1780) with the default sampler, Euler Ancestral as the shipped reference runs it
        (src/samplers.h:1431-1449, ORIGINAL_SAMPLER_ALGORITHMS):  x += ((x - d) / sigma_i) * (sigma_down - sigma_i) + r * sigma_up,
        every operation rounded to float on its own.  init_latent: N(0,1) start (default: numpy stream; the reference draws
        randn_4_w_h(seed % 1000), :1595) -- it is scaled by sigma[0] here as :1611-1612 does; step_noise(i): the ancestral noise of step i.
        sampler="euler": the plain Euler step of the same branch (src/samplers.h:116-126): x += (x - d) / sigma_i * (sigma_{i+1} - sigma_i), no noise."""
        if sampler not in ("euler_a", "euler"):
            raise ValueError("sampler must be 'euler_a' (the reference's default) or 'euler'")
        sig = sigma_schedule(steps, self.log_sigmas)
        rng = np.random.default_rng(seed)     # the reference draws mt19937 normals; any N(0,1) stream is equivalent for the harness
        x0 = rng.standard_normal(latent_shape, dtype=f32) if init_latent is None else np.asarray(init_latent, f32).reshape(latent_shape)
        x = (x0 * f32(sig[0])).astype(f32)
        for i in range(steps):
            den = self.denoise(x, float(sig[i]), cond, uncond)
            if sampler == "euler":
                x = (x + f32(f32(x - den) / f32(sig[i])) * f32(f32(sig[i + 1]) - f32(sig[i]))).astype(f32)
                if on_step:
                    on_step(i, x)
                continue
            sigma_up, sigma_down = self.ancestral_step_scalars(sig[i], sig[i + 1])
            noise = rng.standard_normal(latent_shape, dtype=f32) if step_noise is None else np.asarray(step_noise(i), f32).reshape(latent_shape)
            x = (x + f32(f32(x - den) / f32(sig[i])) * f32(sigma_down - f32(sig[i])) + noise * sigma_up).astype(f32)
            if on_step:
                on_step(i, x)
        return x

    def loop_scalars(self, sig: np.ndarray, sampler: str = "euler_a"):
        """Per-step fp32 scalars of the loop, computed exactly as denoise()/sample() do: c_in, c_out, t (sigma_to_t), sigma_i, d_sigma =
        sigma_down - sigma_i and sigma_up of the Euler-Ancestral update."""
        steps = len(sig) - 1
        c_in, c_out, ts, s_arr, d_sigma, s_up = (np.empty(steps, f32) for _ in range(6))
        for i in range(steps):
            sigma = float(sig[i])
            c_out[i] = f32(-1.0 * sigma)
            c_in[i] = f32(1.0 / np.sqrt(f32(sigma) * f32(sigma) + 1))
            if sigma not in self._t_cache:
                self._t_cache[sigma] = sigma_to_t(sigma, self.log_sigmas)
            ts[i] = f32(self._t_cache[sigma])
            sigma_up, sigma_down = self.ancestral_step_scalars(sig[i], sig[i + 1])
            if sampler == "euler":      # (the Euler step is the ancestral one with sigma_down = sigma_{i+1} and no noise, src/samplers.h:116-126)
                sigma_up, sigma_down = f32(0.0), f32(sig[i + 1])
            s_arr[i] = f32(sig[i])
            d_sigma[i] = f32(sigma_down - f32(sig[i]))
            s_up[i] = sigma_up
        return c_in, c_out, ts, s_arr, d_sigma, s_up

    def sample_device(self, cond, uncond, steps: int = 20, seed: int = 42, latent_shape=(1, 4, 64, 64), guidance: float = 7.0,
                      init_latent: Optional[np.ndarray] = None, step_noise: Optional[Callable[[int], np.ndarray]] = None,
                      sampler: str = "euler_a") -> np.ndarray:
        """sample() with the whole loop enqueued on the GPU (HIP backend only): per step a scaling kernel fills the UNet's input staging,
        the captured pass is launched, and one kernel does eps -> denoised, the CFG combine and the Euler-Ancestral update -- no host
        round trip until the last step.  Same schedule, same random stream, same fp32 operation order as sample(): the two agree bit
        for bit.  cond / uncond: one context each, or lists with one entry per prompt (latent_shape[0] prompts)."""
        if not self.batched:
            raise RuntimeError("sample_device needs the HIP backend (batched=True)")
        n, P = self.names, latent_shape[0]
        conds = list(cond) if isinstance(cond, (list, tuple)) else [cond] * P
        unconds = list(uncond) if isinstance(uncond, (list, tuple)) else [uncond] * P
        sig = sigma_schedule(steps, self.log_sigmas)
        rng = np.random.default_rng(seed)
        x0 = rng.standard_normal(latent_shape, dtype=f32) if init_latent is None else np.asarray(init_latent, f32).reshape(latent_shape)
        x = np.ascontiguousarray(x0 * f32(sig[0]), f32)
        noise = np.empty((steps,) + tuple(latent_shape), f32)
        c_in, c_out, ts, s_arr, d_sigma, s_up = self.loop_scalars(sig, sampler)
        for i in range(steps):
            noise[i] = rng.standard_normal(latent_shape, dtype=f32) if step_noise is None else np.asarray(step_noise(i), f32).reshape(latent_shape)
        key = (id(self.unet), P)
        if self._dev_ready.get(key):
            for p in range(P):      # plan + captured pass exist: only the contexts change between images
                self.unet.hip_set_input(n["ctx"], 2 * p, conds[p])
                self.unet.hip_set_input(n["ctx"], 2 * p + 1, unconds[p])
        else:
            for _ in range(2):      # run() #1 plans and runs eagerly, #2 captures the pass; both leave the contexts resident
                if P > 1:
                    self.denoise(x, float(sig[0]), conds, unconds, guidance)
                else:
                    self.denoise(x, float(sig[0]), conds[0], unconds[0], guidance)
            self._dev_ready[key] = True
        self.last_loop_ms = self.unet.hip_sampler_loop(n["sample"], n["timestep"], n["out"], x, noise, c_in, c_out, ts, s_arr, d_sigma, s_up, guidance)
        return x

    def decode(self, latents: np.ndarray) -> np.ndarray:
        """decoder_solver: latents * 5.48998 -> VAE decoder -> (y + 1) * 127.5 (src/sd.cpp:1174-1256)."""
        z = (latents * f32(5.48998)).astype(f32)
        ys = self._run(self.vae, [{self.names["vae_in"]: z[i:i + 1]} for i in range(z.shape[0])], self.names["vae_out"])
        return ((np.concatenate(ys) + f32(1.0)) * f32(127.5)).astype(f32)

    def decode_tiled(self, latents: np.ndarray, tile: int = 32, names=("latent_sample", "out_image")) -> np.ndarray:
        """sd_tiled_decoder (src/sd.cpp:1258-1346): a VAE graph built for `tile` x `tile` latents is run over overlapping tiles
        (origins 0, 0.75*tile, ... and a last one flush with the border: 0/24/32 for 64-wide latents) and the 8x upscaled tiles are
        blended with linear ramps over the first tile/2 * 8... = 64 output pixels of every non-border edge.  self.vae must be the
        tile-sized decoder.  With the HIP backend all tiles run as ONE batched pass."""
        z = (latents * f32(5.48998)).astype(f32)
        _, _, H, W = z.shape
        step, ramp = (tile * 3) // 4, tile * 2           # 24 latent pixels, 64 output pixels for tile = 32

        def origins(n):
            o, v = [], 0
            while True:
                v = min(v, n - tile)
                o.append(v)
                if v == n - tile:
                    return o
                v += step
        oy, ox = origins(H), origins(W)
        pushes = [{names[0]: np.ascontiguousarray(z[:, :, y:y + tile, x:x + tile])} for y in oy for x in ox]
        outs = self._run(self.vae, pushes, names[1])
        up = outs[0].shape[-1] // tile                   # 8 for the SD decoders
        ramp = ramp * up // 8
        res = np.zeros((1, outs[0].shape[1], H * up, W * up), f32)
        T8 = tile * up
        yy = np.arange(T8, dtype=f32)[:, None]
        xx = np.arange(T8, dtype=f32)[None, :]
        k = 0
        for y in oy:
            for x in ox:
                f = np.ones((T8, T8), f32)
                if y:
                    f = f * np.where(yy < ramp, yy / f32(ramp), f32(1.0))
                if x:
                    f = f * np.where(xx < ramp, xx / f32(ramp), f32(1.0))
                d = res[:, :, y * up:y * up + T8, x * up:x * up + T8]
                res[:, :, y * up:y * up + T8, x * up:x * up + T8] = outs[k] * f + d * (f32(1.0) - f)
                k += 1
        return ((res + f32(1.0)) * f32(127.5)).astype(f32)

    def txt2img(self, cond: np.ndarray, uncond: np.ndarray, steps: int = 20, seed: int = 42, latent_shape=(1, 4, 64, 64)) -> np.ndarray:
        return self.decode(self.sample(cond, uncond, steps, seed, latent_shape))
