"""Dev tool (GPU box): per-candidate cold timings (OSG_TUNE_COLD=1 OSG_TUNE_DUMP=1) of the small-M GEMMs of the 16x16 / 32x32 / 8x8 levels."""
import os, sys
import numpy as np
os.environ.setdefault("OSG_TUNE_COLD", "1")
os.environ.setdefault("OSG_TUNE_DUMP", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu  # noqa: E402

g = osgpu.Gpu(0)
g._ck(g.lib.osg_set_autotune(g.ctx, 1))
rng = np.random.default_rng(0)
f16 = np.float16
for M, K, N in [(512, 1280, 1280), (512, 1280, 3840), (512, 5120, 1280), (2048, 640, 640), (2048, 640, 1920), (2048, 2560, 640), (128, 1280, 1280), (8192, 320, 320), (8192, 320, 960), (8192, 1280, 320)]:
    a = g.to_dev((rng.standard_normal((M, K), dtype=np.float32)).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
    c = g.empty((M, N), f16)
    g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
    g.sync()
    sys.stderr.write("\n")
