"""Dev tool (GPU box): time a few conv/linear shapes under the kernel's debug variants (env OSG_GEMM_DBG/CFG/SPLITS)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu  # noqa: E402

g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
B = 2


def bench(fn, iters=30):
    fn(); fn()
    g.sync()
    g.timer_start()
    for _ in range(iters):
        fn()
    return g.timer_stop() / iters


tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("OSG_GEMM"))
for H, Cin, Cout, k in ([] if os.environ.get('PROBE_GEMM_ONLY') else [(64, 320, 320, 3), (32, 640, 640, 3), (32, 1280, 1280, 3), (16, 1280, 1280, 3), (64, 640, 640, 3)]):
    x = g.to_dev((rng.standard_normal((B, H, H, Cin), dtype=np.float32)).astype(f16))
    w = g.to_dev((rng.standard_normal((Cout, k, k, Cin), dtype=np.float32) * 0.02).astype(f16))
    b = g.to_dev(np.zeros(Cout, f16))
    y = g.empty((B, H, H, Cout), f16)
    def fn():
        g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, x.ptr, w.ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, k, k, 1, 1, k // 2, k // 2, k // 2, k // 2, 0))
    ms = bench(fn)
    fl = 2.0 * B * H * H * Cin * Cout * k * k
    print(f"[{tag}] conv{k}x{k} {H:3d}x{H:<3d} {Cin:5d}->{Cout:<5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s")
for M, K, N in [(8192, 8192, 8192), (4096, 4096, 4096), (8192, 320, 2560), (8192, 1280, 320), (8192, 320, 960), (2048, 640, 5120), (2048, 2560, 640)]:
    a = g.to_dev((rng.standard_normal((M, K), dtype=np.float32)).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
    c = g.empty((M, N), f16)
    def fn():
        g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
    ms = bench(fn, 10)
    got = c.numpy()[:4].astype(np.float64)
    want = a.numpy()[:4].astype(np.float64) @ w.numpy().astype(np.float64).T
    err = np.abs(got - want).max() / np.abs(want).max()
    print(f"[{tag}] gemm M={M} K={K} N={N}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s  relerr(first rows) {err:.1e}")
