"""Dev tool (GPU box): GroupNorm+SiLU -> Conv3x3 fused (osg_group_norm_conv3x3) vs separate launches, per SD1.5 resnet shape."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
B = 2
def bench(fn, it=30):
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(it): fn()
    return g.timer_stop() / it * 1e3
for H, Cin, Cout in [(64, 320, 320), (64, 640, 320), (32, 640, 640), (32, 1280, 640), (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280)]:
    x = g.to_dev(rng.standard_normal((B, H, H, Cin), dtype=np.float32).astype(f16))
    ga = g.to_dev(np.ones(Cin, f16)); be = g.to_dev(np.zeros(Cin, f16))
    w = g.to_dev((rng.standard_normal((Cout, 3, 3, Cin), dtype=np.float32) * 0.02).astype(f16))
    b = g.to_dev(np.zeros(Cout, f16))
    yn = g.empty((B, H, H, Cin), f16); y = g.empty((B, H, H, Cout), f16)
    def sep():
        g._ck(g.lib.osg_group_norm_nhwc(g.ctx, 2, x.ptr, ga.ptr, be.ptr, yn.ptr, B, H * H, Cin, 32, 1e-5, 1))
        g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, yn.ptr, w.ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0))
    def conv_only():
        g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, yn.ptr, w.ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, 3, 3, 1, 1, 1, 1, 1, 1, 0))
    def fused():
        g._ck(g.lib.osg_group_norm_conv3x3(g.ctx, x.ptr, ga.ptr, be.ptr, 32, 1e-5, 1, w.ptr, b.ptr, 2, None, 0, None, y.ptr, B, H, H, Cin, Cout))
    print(f"{H}x{H} {Cin}->{Cout}: conv only {bench(conv_only):7.1f} us | GN+conv separate {bench(sep):7.1f} us | fused {bench(fused):7.1f} us")
