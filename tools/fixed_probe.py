"""Dev tool (GPU box): per-launch fixed cost -- time vs K for the halo conv and the GEMM at tiny K."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
B = 2
def bench(fn, it=50):
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(it): fn()
    return g.timer_stop() / it * 1e3
for H, Cout in [(64, 320), (16, 1280)]:
    for Cin in [64, 128, 320, 640]:
        x = g.to_dev(rng.standard_normal((B, H, H, Cin), dtype=np.float32).astype(f16))
        w = g.to_dev((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * 0.02).astype(f16))
        b = g.to_dev(np.zeros(Cout, f16)); y = g.empty((B, H, H, Cout), f16)
        def fn():
            g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, x.ptr, w.ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0))
        print(f"conv3x3 {H}x{H} {Cin}->{Cout}: {bench(fn):6.1f} us ({Cin//64*9} units)")
for M, N in [(8192, 320), (512, 1280)]:
    for K in [64, 128, 320, 1280]:
        a = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
        w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
        c = g.empty((M, N), f16)
        def fn():
            g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
        print(f"gemm M={M} N={N} K={K}: {bench(fn):6.1f} us ({K//64} k-tiles)")
x = g.to_dev(np.zeros(1024, f16)); y = g.empty((1024,), f16)
def tiny():
    g._ck(g.lib.osg_unary(g.ctx, 2, 5, x.ptr, y.ptr, 1024, 0.0))
print(f"trivial elementwise kernel (1024 elements): {bench(tiny):6.1f} us")
