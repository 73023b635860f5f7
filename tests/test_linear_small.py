"""-m gpu: the lean linear launch (osg_linear_small, onnxstream_amd/csrc/osg_linsmall.hip) -- the transformer blocks' projections and the 1x1 convolutions of the
SD 1.5 UNet at their real shapes -- against osg_gemm on the same operands (same MFMA shape, f32 accumulation in k order: a couple of f16 ulps at most, mostly
the same bits) and against the numpy restatement of the reference arithmetic (oracle/np_ops.py; MatMul + Add + Add, src/onnxstream.cpp:5669-5861, :3906-4000;
with LayerNorm: the chain :5237-5604 in front, against osg_layer_norm + osg_gemm)."""
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


# (M, N, K, LayerNorm): every (tile, K) class the planner routes here
SHAPES = [(8192, 320, 320, False), (8192, 960, 320, True), (2048, 640, 640, False), (2048, 1920, 640, True), (2048, 640, 640, True), (2048, 640, 320, False),
          (512, 1280, 1280, False), (512, 3840, 1280, True), (512, 1280, 1280, True), (128, 1280, 1280, False), (128, 3840, 1280, True), (512, 1280, 640, False),
          (128, 1280, 2560, False), (512, 1280, 2560, False), (512, 1280, 1920, False), (2048, 640, 1920, False), (2048, 640, 1280, False), (2048, 640, 960, False),
          (8192, 320, 960, False), (8192, 320, 640, False), (64, 64, 320, False), (16, 128, 2560, True)]


@pytest.mark.parametrize("M,N,K,ln", SHAPES)
def test_linear_small(gpu, M, N, K, ln):
    assert gpu.lib.osg_linear_small_supported(M, N, K, int(ln)) == 1
    rng = np.random.default_rng(M + 3 * N + 7 * K + ln)
    x, w = rnd(rng, (M, K), 1.5), rnd(rng, (N, K), K ** -0.5)
    if ln:
        x = (x.astype(f32) + 0.3).astype(f16)
    bias, res = rnd(rng, (N,), 0.2), rnd(rng, (M, N))
    gam, bet = (1 + 0.2 * rng.standard_normal(K)).astype(f16), rnd(rng, (K,), 0.1)
    dx, dw, db, dr = gpu.to_dev(x), gpu.to_dev(w), gpu.to_dev(bias), gpu.to_dev(res)
    dwp = gpu.tblock_pack_weight(dw)
    dg, dbt = (gpu.to_dev(gam), gpu.to_dev(bet)) if ln else (None, None)
    got = gpu.linear_small(dx, dwp, db, dr, dg, dbt, 1e-5).numpy()
    assert np.isfinite(got.astype(f32)).all()
    if ln:
        xn_dev = gpu.layer_norm(dx, dg, dbt, 1e-5)
        sep = gpu.gemm(xn_dev, dw, db, dr, b_is_nk=True).numpy()
        xn = ref.layer_norm_exact(x, gam, bet, 1e-5)
    else:
        sep = gpu.gemm(dx, dw, db, dr, b_is_nk=True).numpy()
        xn = x
    want = ref.matmul(xn, w.T, bias, res)
    e_sep, e_np, e_ref = rel_max(got, sep), rel_max(got, want), rel_max(sep, want)
    print(f"M={M} N={N} K={K} ln={ln}: lean-vs-gemm {e_sep:.2e}  lean-vs-numpy {e_np:.2e}  gemm-vs-numpy {e_ref:.2e}")
    assert e_np <= max(1e-3, 1.5 * e_ref) and e_sep <= 1.5e-3
    # no bias, no residual, a second destination with a row pitch
    wide = gpu.to_dev(np.full((M, N + 72), 3.0, f16))
    got2 = gpu.linear_small(dx, dwp, None, None, dg, dbt, 1e-5, out2=wide, out2_col=8)
    g2, wd = got2.numpy(), wide.numpy()
    assert np.array_equal(wd[:, 8:8 + N], g2) and (wd[:, :8] == 3).all() and (wd[:, 8 + N:] == 3).all()
    assert rel_max(g2, ref.matmul(xn, w.T)) <= max(1e-3, 1.5 * rel_max(gpu.gemm(xn_dev if ln else dx, dw, None, None, b_is_nk=True).numpy(), ref.matmul(xn, w.T)))
    assert np.array_equal(gpu.linear_small(dx, dwp, db, dr, dg, dbt, 1e-5).numpy(), got)          # relaunch: same bits
    # the row statistics hand-over to a LayerNorm-folding consumer (osg_gemm_rowstats' format): sums of the ROUNDED outputs per 32-column slot
    if not ln and N % 32 == 0 and gpu.lib.osg_linear_small_rowstats_supported(M, N, K) == 1:
        rs = gpu.to_dev(np.zeros((M, N // 32, 2), f32))
        got3 = gpu.linear_small(dx, dwp, db, dr, rowstats=rs).numpy()
        assert np.array_equal(got3, got)
        g64 = got.astype(np.float64).reshape(M, N // 32, 32)
        r = rs.numpy().astype(np.float64)
        assert np.allclose(r[..., 0], g64.sum(-1), rtol=1e-5, atol=1e-3) and np.allclose(r[..., 1], (g64 * g64).sum(-1), rtol=1e-5, atol=1e-3)


def test_linear_small_rejects_what_it_does_not_take(gpu):
    for M, N, K in [(2048, 640, 768), (100, 640, 640), (2048, 100, 640), (2048, 640, 2880), (154, 640, 640)]:
        assert gpu.lib.osg_linear_small_supported(M, N, K, 0) == 0
