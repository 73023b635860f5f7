"""Dev tool (CPU, needs oracle/_ref): per-op-type mismatch counts of the uint8 restatements (oracle/np_qu8.py) against the reference's own
intermediates on the miniature fully-uint8 VAE.  See oracle/qu8_check.py."""
import os, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.synth import sd_vae
from onnxstream_amd.synth.graph import DirSink
from oracle import qu8_check, ref as oref

with tempfile.TemporaryDirectory() as d:
    d += "/"
    sd_vae.build_vae_decoder(DirSink(d), sd_vae.TINY_VAE, quant_all=True)
    z = np.random.default_rng(77).standard_normal((1, 4, sd_vae.TINY_VAE.latent, sd_vae.TINY_VAE.latent)).astype(np.float32)
    ranges = oref.calibrate_ranges(d, {"input.1": z})
    for t, (n, e, b) in qu8_check.verify(d, {"input.1": z}, ranges).items():
        print(f"{t}: {n} ops, {e} codes, {b} mismatches")
