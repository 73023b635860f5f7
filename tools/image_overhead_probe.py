"""Dev tool (GPU box), round 6: where the time of ONE image goes outside the 20 captured UNet passes, and the per-step breakdown of the fp16 VAE decoder.
bench.py's headline line has ms_per_step - unet_device_ms_per_step = ~0.38 ms: the VAE decode (88 launches) amortised over 20 steps + host work per image.
    python tools/image_overhead_probe.py [out_prefix]      (OSG_TUNE_CACHE honoured; synthetic graphs as bench.py builds them)
"""
import dataclasses
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnxstream_amd import build as b
from onnxstream_amd.pipeline import Txt2Img, sigma_schedule
from onnxstream_amd.synth import sd_unet, sd_vae
from onnxstream_amd.synth.graph import DirSink

out_prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_image_overhead"
root = os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth")
cfg = sd_unet.SD15
vcfg = dataclasses.replace(sd_vae.SD_VAE, latent=cfg.latent, name=f"sd_vae{cfg.latent}")
ud, vd = os.path.join(root, cfg.name) + "/", os.path.join(root, vcfg.name) + "/"
if not os.path.exists(ud + ".complete"):
    os.makedirs(ud, exist_ok=True)
    sd_unet.build_unet(DirSink(ud), cfg)
    open(ud + ".complete", "w").write("ok")
if not os.path.exists(vd + ".complete"):
    os.makedirs(vd, exist_ok=True)
    sd_vae.build_vae_decoder(DirSink(vd), vcfg)
    open(vd + ".complete", "w").write("ok")

pipe = Txt2Img(b.LIB_HOST, ud, vd, batched=True, device=0, autotune=True)
rng = np.random.default_rng(0)
cond = rng.standard_normal((1, 77, cfg.ctx_dim), dtype=np.float32)
uncond = rng.standard_normal((1, 77, cfg.ctx_dim), dtype=np.float32)
STEPS = 20
sig = sigma_schedule(STEPS, pipe.log_sigmas)
lat = (1, 4, cfg.latent, cfg.latent)
x0 = rng.standard_normal(lat, dtype=np.float32)
for _ in range(3):
    pipe.denoise(x0 * sig[0], float(sig[0]), cond, uncond)
    pipe.decode(x0)
for _ in range(2):
    pipe.sample_device(cond, uncond, steps=STEPS, seed=1, latent_shape=lat)

T = lambda: time.perf_counter()
rows = []
for rep in range(6):
    t0 = T()
    x = pipe.sample_device(cond, uncond, steps=STEPS, seed=rep, latent_shape=lat)
    t1 = T()
    loop_dev = pipe.last_loop_ms
    z = (x * np.float32(5.48998)).astype(np.float32)
    t2 = T()
    pipe.vae.add_tensor(pipe.names["vae_in"], z)
    t3 = T()
    pipe.vae.run()
    t4 = T()
    y = pipe.vae.get_tensor(pipe.names["vae_out"], 0)[0]
    t5 = T()
    pipe.vae.clear_tensors()
    img = ((y + np.float32(1.0)) * np.float32(127.5)).astype(np.float32)
    t6 = T()
    rows.append(dict(sample_device_wall=(t1 - t0) * 1e3, loop_device=loop_dev, sample_host_outside_loop=(t1 - t0) * 1e3 - loop_dev, scale=(t2 - t1) * 1e3,
                     vae_add_tensor=(t3 - t2) * 1e3, vae_run_wall=(t4 - t3) * 1e3, vae_device=pipe.vae.hip_last_pass_ms(), vae_get_tensor=(t5 - t4) * 1e3,
                     post=(t6 - t5) * 1e3, image_wall=(t6 - t0) * 1e3))
with open(out_prefix + ".txt", "w") as f:
    f.write("# one image = sample_device (20 steps, one host sync) + VAE decode; ms, 6 repetitions then the median\n")
    keys = list(rows[0].keys())
    f.write("\t".join(keys) + "\n")
    for r in rows:
        f.write("\t".join(f"{r[k]:.3f}" for k in keys) + "\n")
    med = {k: float(np.median([r[k] for r in rows])) for k in keys}
    f.write("median\n" + "\t".join(f"{med[k]:.3f}" for k in keys) + "\n")
    f.write(f"# per step of 20: image_wall {med['image_wall'] / STEPS:.4f}  loop_device {med['loop_device'] / STEPS:.4f}  everything else {(med['image_wall'] - med['loop_device']) / STEPS:.4f}\n")
print(open(out_prefix + ".txt").read())

# ---- the path bench.py times: hip_sampler_loop called directly (schedule scalars precomputed, noise drawn beforehand), then pipe.decode
scal = pipe.loop_scalars(sig)
n = pipe.names
rows2 = []
for rep in range(6):
    x = np.ascontiguousarray(rng.standard_normal(lat, dtype=np.float32) * sig[0], np.float32)
    noise = rng.standard_normal((STEPS,) + lat, dtype=np.float32)
    t0 = T()
    dev = pipe.unet.hip_sampler_loop(n["sample"], n["timestep"], n["out"], x, noise, *scal, 7.0, None)
    t1 = T()
    img = pipe.decode(x)
    t2 = T()
    rows2.append(dict(loop_wall=(t1 - t0) * 1e3, loop_device=dev, loop_host_outside_device=(t1 - t0) * 1e3 - dev, decode_wall=(t2 - t1) * 1e3,
                      vae_device=pipe.vae.hip_last_pass_ms(), image_wall=(t2 - t0) * 1e3))
with open(out_prefix + ".txt", "a") as f:
    f.write("# the path bench.py times: Model.hip_sampler_loop directly + Txt2Img.decode; ms\n")
    keys = list(rows2[0].keys())
    f.write("\t".join(keys) + "\n")
    for r in rows2:
        f.write("\t".join(f"{r[k]:.3f}" for k in keys) + "\n")
    med = {k: float(np.median([r[k] for r in rows2])) for k in keys}
    f.write("median\n" + "\t".join(f"{med[k]:.3f}" for k in keys) + "\n")
print(open(out_prefix + ".txt").read())
if os.environ.get("PROBE_NO_VAE_BREAKDOWN"):
    sys.exit(0)

prof = pipe.vae.hip_profile(3)
by = {}
for ms, fl, byt, what in prof:
    k = what.split(" ", 1)[0].split("+", 1)[0]
    e = by.setdefault(k, [0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += ms; e[2] += fl; e[3] += byt
with open(out_prefix + "_vae_breakdown.txt", "w") as f:
    f.write("# fp16 VAE decoder (64x64 latent -> 512x512 image), eager pass with HIP events around every step, 3 repetitions\n# kind\tlaunches\tms\tGFLOP\tTFLOP/s\tGB\tGB/s\n")
    for k, e in sorted(by.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{k}\t{e[0]}\t{e[1]:.4f}\t{e[2] / 1e9:.2f}\t{e[2] / max(e[1], 1e-9) / 1e9:.1f}\t{e[3] / 1e9:.4f}\t{e[3] / max(e[1], 1e-9) / 1e6:.1f}\n")
    f.write(f"# total {sum(e[1] for e in by.values()):.4f} ms, {sum(e[2] for e in by.values()) / 1e9:.1f} GFLOP\n# per step: ms\tflops\tbytes\twhat\n")
    for ms, fl, byt, what in prof:
        f.write(f"{ms:.5f}\t{fl:.0f}\t{byt:.0f}\t{what}\n")
print(open(out_prefix + "_vae_breakdown.txt").read()[:6000])
