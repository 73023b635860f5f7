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


@pytest.mark.skipif(not oref.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("shape,threads", [((1, 4, 64, 64), 1), ((1, 4, 64, 64), 4), ((1, 4, 128, 128), 7), ((1, 3, 7, 5), 4)])
def test_dynamic_input_quantisation_matches_push_tensor(shape, threads):
    """What push_tensor does to a pushed fp32 input under uint8 arithmetic (0.1 % percentiles per 16 K chunk per worker thread -> scale /
    zero point -> codes): the restatement reproduces the reference's codes AND its scale bit for bit, for several thread counts."""
    import ctypes
    from onnxstream_amd.bindings import Model
    from oracle import np_qu8 as Q
    z = (np.random.default_rng(sum(shape) + threads).standard_normal(shape) * 3).astype(np.float32)
    m = Model(oref.REF_LIB, threads, "ram+nocache")
    oref._extra(m.lib)
    lib = m.lib
    lib.ref_get_tensor_any.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_void_p)]
    lib.ref_get_tensor_any.restype = ctypes.c_size_t
    m._set_option("use_uint8_arithmetic", 1)
    dims = (ctypes.c_uint * z.ndim)(*z.shape)
    assert not lib.ref_push_tensor_f32(m.handle, b"x", z.ndim, dims, z.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    dt, rank, scale, zp, ptr = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_float(), ctypes.c_int(), ctypes.c_void_p()
    shp = (ctypes.c_size_t * 8)()
    n = lib.ref_get_tensor_any(m.handle, b"x", ctypes.byref(dt), ctypes.byref(rank), shp, ctypes.byref(scale), ctypes.byref(zp), ctypes.byref(ptr))
    ref_codes = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)).copy()
    m.close()
    q, s, z0 = Q.quantize_dynamic(z, threads=threads)
    assert np.float32(s) == np.float32(scale.value) and z0 == zp.value
    assert np.array_equal(q.ravel(), ref_codes)
