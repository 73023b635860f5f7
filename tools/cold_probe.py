"""Dev tool (GPU box): hot vs cold operands.  The same GEMM / conv launched (a) on ONE buffer set over and over (operands L2 / Infinity-Cache
resident: what an isolated microbenchmark sees) and (b) cycling through enough buffer sets to exceed the 256 MiB Infinity Cache, each launch
reading activations and weights nobody touched recently (what a launch inside a UNet pass sees), plus (c) a dependent chain C_i -> A_{i+1}."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
def timeit(fns, it):
    for f in fns[:2]: f()
    g.sync(); g.timer_start()
    for i in range(it): fns[i % len(fns)]()
    return g.timer_stop() / it * 1e3
for (M, N, K) in [(8192, 320, 320), (8192, 320, 1280), (8192, 2560, 320), (2048, 640, 640), (512, 1280, 1280), (128, 1280, 1280)]:
    per = (M * K + N * K + M * N) * 2
    nset = max(2, min(96, int(600e6 // per) + 1))
    sets = []
    for s in range(nset):
        a = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
        w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
        c = g.empty((M, N), f16)
        sets.append((a, w, c))
    def mk(a, w, c):
        return lambda: g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
    hot = timeit([mk(*sets[0])], 200)
    cold = timeit([mk(*s) for s in sets], 200)
    # cold weights only (same activations)
    coldw = timeit([mk(sets[0][0], s[1], sets[0][2]) for s in sets], 200)
    print(f"gemm {M}x{N}x{K}: hot {hot:6.1f} us   cold weights {coldw:6.1f} us   cold everything ({nset} sets) {cold:6.1f} us", flush=True)
B = 2
for (H, Cin, Cout) in [(64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280)]:
    per = (B * H * H * (Cin + Cout) + 9 * Cin * Cout) * 2
    nset = max(2, min(64, int(600e6 // per) + 1))
    sets = []
    for s in range(nset):
        x = g.to_dev(rng.standard_normal((B, H, H, Cin), dtype=np.float32).astype(f16))
        w = g.to_dev((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * 0.02).astype(f16))
        y = g.empty((B, H, H, Cout), f16)
        sets.append((x, w, y))
    def mk(x, w, y):
        return lambda: g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, x.ptr, w.ptr, None, 2, None, y.ptr, B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0))
    hot = timeit([mk(*sets[0])], 100)
    cold = timeit([mk(*s) for s in sets], 100)
    coldw = timeit([mk(sets[0][0], s[1], sets[0][2]) for s in sets], 100)
    print(f"conv3x3 {H}x{H} {Cin}->{Cout}: hot {hot:6.1f} us   cold weights {coldw:6.1f} us   cold everything ({nset} sets) {cold:6.1f} us", flush=True)
