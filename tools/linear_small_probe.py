"""Dev probe (GPU box): osg_linear_small against osg_gemm (gemm2_kernel, cost-model choice) on the SD 1.5 pass's projection / 1x1 convolution shapes, cycling
through enough weight copies that every launch finds its weights cold."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
SH = [(8192, 320, 320, 0), (8192, 960, 320, 1), (2048, 640, 640, 0), (2048, 1920, 640, 1), (2048, 640, 640, 1), (2048, 640, 320, 0), (512, 1280, 1280, 0), (512, 3840, 1280, 1),
      (512, 1280, 1280, 1), (128, 1280, 1280, 0), (512, 1280, 640, 0), (128, 1280, 2560, 0), (512, 1280, 2560, 0), (2048, 640, 1920, 0), (2048, 640, 1280, 0), (8192, 320, 960, 0)]
for M, N, K, ln in SH:
    ncopy = max(4, min(64, int(300e6 / (N * K * 2))))
    x = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
    ws = [g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * K ** -0.5).astype(f16)) for _ in range(ncopy)]
    wps = [g.tblock_pack_weight(w) for w in ws]
    b = g.to_dev(np.zeros(N, f16)); r = g.to_dev(rng.standard_normal((M, N), dtype=np.float32).astype(f16))
    ga = g.to_dev(np.ones(K, f16)); be = g.to_dev(np.zeros(K, f16))
    y = g.empty((M, N), f16); xn = g.empty((M, K), f16)
    def lean(i):
        g._ck(g.lib.osg_linear_small(g.ctx, x.ptr, K, wps[i % ncopy].ptr, b.ptr, r.ptr, N, ga.ptr if ln else None, be.ptr if ln else None, 1e-5, y.ptr, N, None, 0, M, N, K, None))
    def gemm(i):
        if ln:
            g._ck(g.lib.osg_layer_norm(g.ctx, 2, x.ptr, ga.ptr, be.ptr, xn.ptr, M, K, 1e-5))
        g._ck(g.lib.osg_gemm(g.ctx, 2, (xn if ln else x).ptr, ws[i % ncopy].ptr, 1, b.ptr, 2, r.ptr, y.ptr, M, N, K, 1, 0, 0, 0, 0))
    def bench(fn, it=3 * ncopy):
        for i in range(4): fn(i)
        g.sync(); g.timer_start()
        for i in range(it): fn(i)
        return g.timer_stop() / it * 1e3
    tl, tg = bench(lean), bench(gemm)
    print(f"M={M:5d} N={N:5d} K={K:5d} ln={ln}: lean {tl:6.1f} us | osg_gemm{' + LayerNorm launch' if ln else ''} {tg:6.1f} us   ({2.0 * M * N * K / 1e9:.2f} GFLOP, {ncopy} weight copies)")
