"""Dev tool (GPU box): one conv shape under env-selected kernel variants."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
B = 2
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("OSG_"))
shapes = [(64, 320, 320), (64, 640, 640), (32, 640, 640), (16, 1280, 1280)]
if os.environ.get("PROBE_SHAPE"):
    shapes = [shapes[int(os.environ["PROBE_SHAPE"])]]
for H, Cin, Cout in shapes:
    x = g.to_dev(rng.standard_normal((B, H, H, Cin), dtype=np.float32).astype(f16))
    w = g.to_dev((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * 0.02).astype(f16))
    b = g.to_dev(np.zeros(Cout, f16))
    y = g.empty((B, H, H, Cout), f16)
    def fn():
        g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, x.ptr, w.ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0))
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(30): fn()
    ms = g.timer_stop() / 30
    print(f"[{tag}] conv3x3 {H}x{H} {Cin}->{Cout}: {ms*1e3:7.1f} us {2.0*B*H*H*Cin*Cout*9/ms/1e9:7.1f} TF/s")
