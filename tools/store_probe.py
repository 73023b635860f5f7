"""Dev tool (GPU box): is the per-launch fixed cost of the M = 8192 GEMMs their OUTPUT? time vs N at one k-tile, and vs M."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
def bench(fn, it=100):
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(it): fn()
    return g.timer_stop() / it * 1e3
for M in (8192, 2048, 512):
    for N in (64, 128, 320, 640, 1280, 2560):
        for K in (64, 320):
            a = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
            w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
            c = g.empty((M, N), f16)
            def fn():
                g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
            t = bench(fn)
            print(f"gemm M={M} N={N} K={K}: {t:6.1f} us   out {M*N*2/1e6:6.2f} MB -> {M*N*2/1e6/t*1e6/1e6:5.2f} TB/s of output", flush=True)
# plain copy kernels for reference: D2D memcpy of the same sizes
for mb in (1, 5, 10, 21, 42):
    n = mb * 1000000 // 2
    x = g.to_dev(np.zeros(n, f16)); y = g.empty((n,), f16)
    def cp():
        g._ck(g.lib.osg_copy(g.ctx, y.ptr, x.ptr, n * 2))
    print(f"osg_copy {mb} MB: {bench(cp):6.1f} us")
    def un():
        g._ck(g.lib.osg_unary(g.ctx, 2, 5, x.ptr, y.ptr, n, 0.0))
    print(f"unary neg {mb} MB: {bench(un):6.1f} us")
