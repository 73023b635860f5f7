"""Dev tool (GPU box): LayerNorm + GEMM as two launches vs osg_gemm_ln (LayerNorm folded), per SD1.5 transformer shape."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
if os.environ.get("TUNE", "1") == "1":
    g.lib.osg_set_autotune(g.ctx, 1)
rng = np.random.default_rng(0)
f16 = np.float16
def timeit(fn, n=40):
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(n): fn()
    return g.timer_stop() / n * 1e3
for (M, K, N, act) in [(8192, 320, 960, 0), (8192, 320, 320, 0), (8192, 320, 2560, 3), (2048, 640, 1920, 0), (2048, 640, 640, 0), (2048, 640, 5120, 3),
                       (512, 1280, 3840, 0), (512, 1280, 1280, 0), (512, 1280, 10240, 3)]:
    x = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * K ** -0.5).astype(f16))
    ga = g.to_dev(np.ones(K, f16)); be = g.to_dev(np.zeros(K, f16)); bias = g.to_dev(np.zeros(N, f16))
    c1 = g.to_dev(np.zeros(N, np.float32)); c2 = g.to_dev(np.zeros(N, np.float32))
    xn = g.empty((M, K), f16)
    y = g.empty((M, N // 2 if act == 3 else N), f16)
    def ln(): g._ck(g.lib.osg_layer_norm(g.ctx, 2, x.ptr, ga.ptr, be.ptr, xn.ptr, M, K, 1e-5))
    def gemm(): g._ck(g.lib.osg_gemm(g.ctx, 2, xn.ptr, w.ptr, 1, bias.ptr, 2, None, y.ptr, M, N, K, 1, 0, 0, 0, act))
    def both(): ln(); gemm()
    def fold(): g._ck(g.lib.osg_gemm_ln(g.ctx, x.ptr, w.ptr, c1.ptr, c2.ptr, 1e-5, None, None, y.ptr, M, N, K, act))
    rs = g.to_dev(np.ones((M, K // 32, 2), np.float32))
    def fold_rs(): g._ck(g.lib.osg_gemm_ln(g.ctx, x.ptr, w.ptr, c1.ptr, c2.ptr, 1e-5, rs.ptr, None, y.ptr, M, N, K, act))
    t_ln, t_g, t_b, t_f, t_r = timeit(ln), timeit(gemm), timeit(both), timeit(fold), timeit(fold_rs)
    print(f"M={M} K={K} N={N} act={act}:  LN {t_ln:5.1f}  GEMM {t_g:5.1f}  LN+GEMM {t_b:5.1f}  folded(in-loop) {t_f:5.1f}  folded(rowstats) {t_r:5.1f} us")
for (M, K) in [(8192, 320), (2048, 640), (512, 1280)]:
    a = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16)); res = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
    w = g.to_dev((rng.standard_normal((K, K), dtype=np.float32) * K ** -0.5).astype(f16)); bias = g.to_dev(np.zeros(K, f16))
    y = g.empty((M, K), f16); rs = g.empty((M, K // 32, 2), np.dtype(np.float32))
    def plain(): g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, bias.ptr, 2, res.ptr, y.ptr, M, K, K, 1, 0, 0, 0, 0))
    def withrs(): g._ck(g.lib.osg_gemm_rowstats(g.ctx, a.ptr, w.ptr, bias.ptr, 2, res.ptr, y.ptr, M, K, K, 0, rs.ptr))
    print(f"producer M={M} K=N={K}: plain {timeit(plain):5.1f}  +rowstats {timeit(withrs):5.1f} us")
