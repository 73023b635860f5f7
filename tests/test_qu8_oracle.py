"""CPU: the numpy restatement of the reference's uint8-arithmetic ops (oracle/np_qu8.py -- Conv, MatMul, Add, Mul, InstanceNormalization,
Sigmoid, Softmax, Resize, Reshape, Transpose as m_use_uint8_arithmetic computes them) is pinned against the REFERENCE ITSELF: every op
output of the miniature fully-uint8 VAE decoder, kept raw inside oracle/_ref, must be reproduced code for code from the op's inputs.
(Groundwork: the HIP backend has no uint8 activations yet -- this is the specification its kernels will be written against.)"""
import tempfile

import numpy as np
import pytest

from onnxstream_amd.synth import sd_vae
from onnxstream_amd.synth.graph import DirSink
from oracle import ref as oref


@pytest.mark.skipif(not oref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_uint8_restatements_reproduce_every_reference_intermediate():
    from oracle import qu8_check
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_vae.build_vae_decoder(DirSink(d), sd_vae.TINY_VAE, quant_all=True)
        z = np.random.default_rng(77).standard_normal((1, 4, sd_vae.TINY_VAE.latent, sd_vae.TINY_VAE.latent)).astype(np.float32)
        ranges = oref.calibrate_ranges(d, {"input.1": z})
        stats = qu8_check.verify(d, {"input.1": z}, ranges)
    assert set(stats) == {"Conv", "Reshape", "InstanceNormalization", "Mul", "Add", "Sigmoid", "Transpose", "MatMul", "Softmax", "Resize"}
    assert sum(v[0] for v in stats.values()) >= 173            # every op but the first convolution (its pushed input is consumed)
    for t, (n_ops, codes, bad) in stats.items():
        assert bad == 0, (t, n_ops, codes, bad)


def test_uint8_add_multiplier_construction_matches_known_case():
    """XNNPACK's qu8 add: 20-bit multiplier for the larger scale ratio, rounding folded into the bias -- one hand-checked case."""
    from oracle import np_qu8 as Q
    a = np.asarray([0, 10, 128, 255], np.uint8)
    b = np.asarray([255, 10, 128, 0], np.uint8)
    out = Q.add_u8(a, 0.5, 128, b, 0.25, 128, 1.0, 128)
    # (a-128)*0.5 + (b-128)*0.25 = [-32.25, -88.5, 0, 31.5]; rounded half up in the fixed-point domain, + 128
    assert out.tolist() == [96, 40, 128, 160]
