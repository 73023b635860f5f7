"""Dev tool (GPU box): the GEGLU projection GEMMs of the SD1.5 transformer blocks (Linear + bias + GEGLU epilogue)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
g.lib.osg_set_autotune(g.ctx, 1)
rng = np.random.default_rng(0)
f16 = np.float16
for (M, K, N) in [(8192, 320, 2560), (2048, 640, 5120), (512, 1280, 10240), (128, 1280, 10240)]:
    x = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * K ** -0.5).astype(f16))
    bias = g.to_dev(np.zeros(N, f16))
    y = g.empty((M, N // 2), f16); y2 = g.empty((M, N), f16)
    def geglu(): g._ck(g.lib.osg_gemm(g.ctx, 2, x.ptr, w.ptr, 1, bias.ptr, 2, None, y.ptr, M, N, K, 1, 0, 0, 0, 3))
    def plain(): g._ck(g.lib.osg_gemm(g.ctx, 2, x.ptr, w.ptr, 1, bias.ptr, 2, None, y2.ptr, M, N, K, 1, 0, 0, 0, 0))
    def t(fn, n=40):
        fn(); fn(); g.sync(); g.timer_start()
        for _ in range(n): fn()
        return g.timer_stop() / n * 1e3
    print(f"M={M} K={K} N={N}: GEGLU epilogue {t(geglu):6.1f} us   plain epilogue (2x the output bytes) {t(plain):6.1f} us")
