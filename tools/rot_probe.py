"""Dev tool (GPU box): GEMM rates with the k-loop rotation experiment (OSG_GEMM_ROT) -- run twice, with and without the variable."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
def bench(fn, it=50):
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(it): fn()
    return g.timer_stop() / it * 1e3
for (M, N, K) in [(8192, 8192, 8192), (8192, 2560, 320), (8192, 320, 1280), (8192, 960, 320), (8192, 320, 320), (2048, 640, 640), (2048, 5120, 640), (2048, 640, 2560), (512, 1280, 1280), (512, 10240, 1280), (512, 1280, 5120)]:
    a = g.to_dev(rng.standard_normal((M, K), dtype=np.float32).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
    c = g.empty((M, N), f16)
    def fn():
        g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
    t = bench(fn)
    print(f"ROT={os.environ.get('OSG_GEMM_ROT','0')} gemm {M}x{N}x{K}: {t:8.1f} us  {2.0*M*N*K/t/1e6:7.1f} TFLOP/s", flush=True)
