"""Generate tests/golden/vae_tiny_qu8.npz from the REFERENCE: the miniature VAE decoder as a FULLY uint8 model (synth quant_all = the
exporter's vae_decoder_qu8 layout) run with m_use_uint8_arithmetic after an fp32 calibration pass -- the W8A8 configuration of
BASELINE.json config 3 as the reference itself can run it (src/sd.cpp:1212-1222).  The HIP backend does not implement uint8 activations
yet; the fixture pins the oracle (incl. the shim's qu8 softmax) for the round that builds them.

    python tools/make_golden_qu8.py        # needs oracle/_ref/libonnxstream_ref.so (make -C oracle ref)"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.synth import sd_vae  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402
from oracle import ref as oref  # noqa: E402

assert oref.available(), "build the oracle first: make -C oracle ref"
with tempfile.TemporaryDirectory() as d:
    d += "/"
    sd_vae.build_vae_decoder(DirSink(d), sd_vae.TINY_VAE, quant_all=True)
    z = np.random.default_rng(77).standard_normal((1, 4, sd_vae.TINY_VAE.latent, sd_vae.TINY_VAE.latent)).astype(np.float32)
    ins = {"input.1": z}
    ranges = oref.calibrate_ranges(d, ins)
    o32 = oref.run_model(d, ins, fp16=False, threads=1)["out_image"]
    o8 = oref.run_model_u8(d, ins, ranges)["out_image"]
    mx = float(np.abs(o32).max())
    print(f"vae_tiny_qu8: {len(ranges.splitlines())} ranges, out {o8.shape}, |u8 - fp32|/max = {np.abs(o8 - o32).max() / mx:.3f}")
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "vae_tiny_qu8.npz"), z=z, ranges=np.asarray(ranges), ref_u8=o8, ref32=o32)
