"""Dev tool (GPU box): GroupNorm(+SiLU) launch time per SD1.5 shape, single-launch slab kernel vs the three-pass path."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
for (N, H, C) in [(2, 64, 320), (2, 64, 640), (2, 64, 960), (2, 32, 640), (2, 32, 960), (2, 32, 1280), (2, 32, 1920), (2, 16, 1280), (2, 16, 1920), (2, 16, 2560), (2, 8, 1280), (2, 8, 2560)]:
    x = g.to_dev(rng.standard_normal((N, H, H, C), dtype=np.float32).astype(f16))
    ga = g.to_dev(np.ones(C, f16)); be = g.to_dev(np.zeros(C, f16))
    y = g.empty((N, H, H, C), f16)
    def fn():
        g._ck(g.lib.osg_group_norm_nhwc(g.ctx, 2, x.ptr, ga.ptr, be.ptr, y.ptr, N, H * H, C, 32, 1e-5, 1))
    res = []
    for off in (False, True):
        if off: os.environ["OSG_GN_SLAB_OFF"] = "1"
        else: os.environ.pop("OSG_GN_SLAB_OFF", None)
        fn(); fn(); g.sync(); g.timer_start()
        for _ in range(50): fn()
        res.append(g.timer_stop() / 50 * 1e3)
    mb = 2 * N * H * H * C * 2 / 1e6
    print(f"GN {N}x{H}x{H}x{C}: slab {res[0]:7.1f} us   three-pass {res[1]:7.1f} us   ({mb:.1f} MB r+w)")
