"""Dev tool (GPU box): the weight-streaming-bound 3x3 convolutions of the 8x8 / 16x16 levels under forced (BN, splits), with the weights
warm (same tensor every launch: MALL-resident) and cold (cycling through > 256 MB of distinct weight tensors, like a real pass)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
B = 2
shapes = [(8, 1280, 1280), (8, 2560, 1280), (16, 1280, 1280), (16, 2560, 1280), (16, 640, 1280), (32, 1280, 640)]
if os.environ.get("PROBE_SHAPE"):
    shapes = [shapes[int(i)] for i in os.environ["PROBE_SHAPE"].split(",")]
cfgs = [None] + [(bn, s) for bn in (80, 128, 160) for s in (4, 5, 8, 10, 20, 40)]
for H, Cin, Cout in shapes:
    wbytes = Cout * 9 * Cin * 2
    ncold = max(2, int(300e6 // wbytes) + 1)
    w0 = (rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * 0.02).astype(f16)
    ws = [g.to_dev(w0) for _ in range(ncold)]
    x = g.to_dev(rng.standard_normal((B, H, H, Cin), dtype=np.float32).astype(f16))
    b = g.to_dev(np.zeros(Cout, f16))
    y = g.empty((B, H, H, Cout), f16)
    print(f"== conv3x3 {H}x{H} {Cin}->{Cout}: weights {wbytes/1e6:.1f} MB, {ncold} cold copies")
    for cfg in cfgs:
        if cfg is None:
            os.environ.pop("OSG_CONV3X3_BN", None); os.environ.pop("OSG_CONV3X3_SPLITS", None)
        else:
            if Cout % cfg[0] and cfg[0] != 128: continue
            if cfg[1] > Cin // 64: continue
            os.environ["OSG_CONV3X3_BN"] = str(cfg[0]); os.environ["OSG_CONV3X3_SPLITS"] = str(cfg[1])
        def fn(i):
            g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, x.ptr, ws[i % ncold].ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0))
        out = []
        for cold in (False, True):
            for i in range(3): fn(i if cold else 0)
            g.sync(); g.timer_start()
            n = 3 * ncold
            for i in range(n): fn(i if cold else 0)
            out.append(g.timer_stop() / n * 1e3)
        print(f"   {'model' if cfg is None else 'BN=%d S=%d' % cfg:14s} warm {out[0]:6.1f} us  cold {out[1]:6.1f} us  ({wbytes/out[1]/1e6:5.2f} TB/s)")
