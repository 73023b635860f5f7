"""Generate tests/golden/*.npz from the REFERENCE (oracle/_ref = unmodified /root/reference sources + XNNPACK).

    python tools/make_golden.py            # needs oracle/_ref/libonnxstream_ref.so (make -C oracle ref)

Each fixture holds: the fp32 inputs, the reference output with fp16 arithmetic (`ref16`, the parity target) and with fp32
arithmetic (`ref32`, used to triangulate: being closer to fp32 than the fp16 reference is not an error)."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_cases as gc  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402
from oracle import ref as oref  # noqa: E402

out_dir = os.path.join(REPO, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)
assert oref.available(), "build the oracle first: make -C oracle ref"
only = sys.argv[1:]          # optional: the cases to (re)generate; default all + the pipeline fixture
for name in (only or gc.all_case_names()):
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        ins = gc.emit(gc.by_name(name), DirSink(d))
        o16 = oref.run_model(d, ins, fp16=True, threads=1)
        o32 = oref.run_model(d, ins, fp16=False, threads=1)
        assert len(o16) == 1, (name, list(o16))
        (oname, v16), = o16.items()
        v32 = o32[oname]
        mx = float(np.abs(v32).max())
        print(f"{name:20s} out={oname} shape={v16.shape} max|ref32|={mx:.3f} |ref16-ref32|/max={np.abs(v16 - v32).max() / mx:.2e}")
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), out_name=np.asarray(oname), ref16=v16, ref32=v32,
                            **{"in_" + k: v for k, v in ins.items()})

if only:
    sys.exit(0)
# ---- end-to-end: 3 Euler-A steps with CFG + VAE decode, driven through the reference library by the same harness the product uses
from onnxstream_amd.pipeline import Txt2Img  # noqa: E402
from onnxstream_amd.synth import sd_unet, sd_vae  # noqa: E402
with tempfile.TemporaryDirectory() as d:
    du, dv = d + "/unet/", d + "/vae/"
    sd_unet.build_unet(DirSink(du), sd_unet.TINY)
    sd_vae.build_vae_decoder(DirSink(dv), sd_vae.TINY_VAE)
    rng = np.random.default_rng(5)
    cond = rng.standard_normal((1, 11, 48), dtype=np.float32)
    uncond = rng.standard_normal((1, 11, 48), dtype=np.float32)
    p = Txt2Img(oref.REF_LIB, du, dv, batched=False, threads=1)
    lat = p.sample(cond, uncond, steps=3, seed=9, latent_shape=(1, 4, 16, 16))
    img = p.decode(lat)
    p.close()
    print(f"pipeline_tiny latents max {np.abs(lat).max():.3f} image range [{img.min():.1f}, {img.max():.1f}]")
    # tiled decode (sd_tiled_decoder): an 8x8-latent decoder over the 16x16 latents, 3x3 overlapping tiles, blended
    import dataclasses
    dt = d + "/vae_t/"
    sd_vae.build_vae_decoder(DirSink(dt), dataclasses.replace(sd_vae.TINY_VAE, latent=8, in_name="latent_sample"))
    pt = Txt2Img(oref.REF_LIB, du, dt, batched=False, threads=1)
    img_t = pt.decode_tiled(lat, tile=8)
    pt.close()
    np.savez_compressed(os.path.join(out_dir, "pipeline_tiny.npz"), cond=cond, uncond=uncond, latents=lat, image=img, image_tiled=img_t)
