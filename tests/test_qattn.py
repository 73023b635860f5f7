"""-m gpu: cross-attention with its query projection inside (osg_qattn, onnxstream_amd/csrc/osg_tchain.hip) against
  (1) the same three ops as separate launches (osg_layer_norm, osg_gemm, osg_attention_strided): q within an f16 ulp of the separate projection's
      (different k order: four k slices added up), the output within 1e-3 of max|want|;
  (2) the numpy restatement of the reference arithmetic (oracle/np_ops.py).
Reference ops restated: the LayerNorm chain src/onnxstream.cpp:5237-5604, MatMul :5669-5861, AttentionFusedOps :6696-6929."""
import numpy as np
import pytest

from oracle import np_ops as ref

pytestmark = pytest.mark.gpu
f16, f32 = np.float16, np.float32


def rel_max(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def rnd(rng, shape, std=1.0):
    return (rng.standard_normal(shape, dtype=f32) * std).astype(f16)


@pytest.mark.parametrize("M,imgs,C,Tk,bias,heads", [(2048, 2, 640, 77, False, 8), (512, 2, 1280, 77, False, 8), (128, 2, 1280, 77, True, 8), (96, 3, 640, 50, True, 8),
                                                     (64, 1, 640, 80, False, 8), (32, 1, 1280, 1, False, 8), (1024, 2, 640, 77, False, 10), (512, 2, 1280, 77, True, 20), (1600, 2, 640, 77, False, 10)])
def test_qattn_against_separate_launches_and_numpy(gpu, M, imgs, C, Tk, bias, heads):
    D = C // heads
    assert gpu.lib.osg_qattn_supported(M, M // imgs, C, heads, Tk) == 1
    rng = np.random.default_rng(M + C + Tk)
    x = rnd(rng, (M, C), 1.5)
    gamma, beta = (1 + 0.2 * rng.standard_normal(C)).astype(f16), rnd(rng, (C,), 0.1)
    wq = rnd(rng, (C, C), C ** -0.5)                      # [N, K]
    bq = rnd(rng, (C,), 0.1) if bias else None
    k, v = rnd(rng, (imgs, Tk, C)), rnd(rng, (imgs, Tk, C))
    scale, eps = D ** -0.5, 1e-5
    dx, dg, db, dk, dv = gpu.to_dev(x), gpu.to_dev(gamma), gpu.to_dev(beta), gpu.to_dev(k), gpu.to_dev(v)
    dwq = gpu.to_dev(wq)
    kp, vtp = gpu.tblock_kv_pack(dk, dv, heads)
    out, qd = gpu.qattn(dx, dg, db, gpu.tblock_pack_weight(dwq), kp, vtp, Tk, heads, scale, M // imgs, eps, bq=gpu.to_dev(bq) if bias else None, debug=True)
    got, got_q = out.numpy(), qd.numpy()
    # separate launches
    n2 = gpu.layer_norm(dx, dg, db, eps)
    q = gpu.gemm(n2, dwq, gpu.to_dev(bq) if bias else None, None, b_is_nk=True)
    sep_q = q.numpy()
    a2 = gpu.attention_tokens(gpu.to_dev(sep_q.reshape(imgs, M // imgs, C)), dk, dv, heads, scale)
    sep = a2.numpy().reshape(M, C)
    # numpy
    n2n = ref.layer_norm_exact(x, gamma, beta, eps)
    qn = ref.matmul(n2n, wq.T, bq) if bias else ref.matmul(n2n, wq.T)
    sp = lambda t, T: t.reshape(imgs, T, heads, D).transpose(0, 2, 1, 3).reshape(imgs * heads, T, D)
    Tq = M // imgs
    want = ref.attention_exact(sp(qn, Tq), sp(k, Tk), sp(v, Tk), scale).reshape(imgs, heads, Tq, D).transpose(0, 2, 1, 3).reshape(M, C)
    msg = (f"q fused-vs-separate {rel_max(got_q, sep_q):.2e} fused-vs-numpy {rel_max(got_q, qn):.2e} separate-vs-numpy {rel_max(sep_q, qn):.2e}; "
           f"out fused-vs-separate {rel_max(got, sep):.2e} fused-vs-numpy {rel_max(got, want):.2e} separate-vs-numpy {rel_max(sep, want):.2e}")
    print("\n" + msg)
    assert np.isfinite(got.astype(f32)).all(), msg
    assert rel_max(got_q, qn) <= max(1e-3, 1.5 * rel_max(sep_q, qn)), msg
    assert rel_max(got, want) <= max(1.5e-3, 1.5 * rel_max(sep, want)), msg
    assert rel_max(got_q, sep_q) <= 1e-3, msg
    # relaunch: same bits, also behind a large fill that evicts the operands from the caches (a missing wait in front of a workgroup barrier showed only on cold operands)
    scratch = gpu.empty((64 * 1024 * 1024,), f16)
    for fillv in (1, 2, 3):
        gpu._ck(gpu.lib.osg_memset(gpu.ctx, scratch.ptr, fillv, 128 * 1024 * 1024))
        again, _ = gpu.qattn(dx, dg, db, gpu.tblock_pack_weight(dwq), kp, vtp, Tk, heads, scale, M // imgs, eps, bq=gpu.to_dev(bq) if bias else None)
        assert np.array_equal(again.numpy(), got), "not reproducible on cold operands"
    out2, _ = gpu.qattn(dx, dg, db, gpu.tblock_pack_weight(dwq), kp, vtp, Tk, heads, scale, M // imgs, eps, bq=gpu.to_dev(bq) if bias else None)
    assert np.array_equal(out2.numpy(), got)


def test_qattn_rejects_what_it_does_not_take(gpu):
    assert gpu.lib.osg_qattn_supported(8192, 4096, 320, 8, 77) == 0      # (the tail kernel's level)
    assert gpu.lib.osg_qattn_supported(2048, 1024, 640, 5, 77) == 0
    assert gpu.lib.osg_qattn_supported(2048, 1024, 640, 20, 77) == 0 and gpu.lib.osg_qattn_supported(2048, 1024, 1280, 10, 77) == 0
    assert gpu.lib.osg_qattn_supported(100, 100, 640, 8, 77) == 0
    assert gpu.lib.osg_qattn_supported(64, 64, 640, 8, 81) == 0
    assert gpu.lib.osg_qattn_supported(8192, 4096, 640, 10, 77) == 0      # (SDXL's 64 x 64 level: more workgroups than the design pays for)
