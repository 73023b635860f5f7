"""Dev tool (GPU box): per-shape throughput of the implicit-GEMM kernel on the SD1.5 UNet layer shapes (batch 2)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu  # noqa: E402

g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16


def bench(fn, iters=20):
    fn(); fn()
    g.sync()
    g.timer_start()
    for _ in range(iters):
        fn()
    return g.timer_stop() / iters


B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
convs = [  # (H, Cin, Cout, k, count)
    (64, 320, 320, 3, 7), (64, 640, 320, 3, 2), (64, 960, 320, 3, 1), (32, 320, 640, 3, 1), (32, 640, 640, 3, 6), (32, 1280, 640, 3, 1),
    (32, 1920, 640, 3, 1), (32, 960, 640, 3, 1), (16, 640, 1280, 3, 1), (16, 1280, 1280, 3, 6), (16, 2560, 1280, 3, 2), (16, 1920, 1280, 3, 1),
    (8, 1280, 1280, 3, 11), (8, 2560, 1280, 3, 3), (16, 1280, 1280, 3, 1), (32, 1280, 1280, 3, 1), (64, 640, 640, 3, 1),
    (64, 320, 320, 1, 10), (32, 640, 640, 1, 10), (16, 1280, 1280, 1, 12)]
tot_t = tot_f = 0
print(f"batch {B}")
for H, Cin, Cout, k, cnt in convs:
    x = g.to_dev((rng.standard_normal((B, H, H, Cin), dtype=np.float32)).astype(f16))
    w = g.to_dev((rng.standard_normal((Cout, k, k, Cin), dtype=np.float32) * 0.02).astype(f16))
    b = g.to_dev(np.zeros(Cout, f16))
    y = g.empty((B, H, H, Cout), f16)
    def fn():
        g._ck(g.lib.osg_conv2d_nhwc(g.ctx, 2, x.ptr, w.ptr, b.ptr, 2, None, y.ptr, B, H, H, Cin, Cout, k, k, 1, 1, k // 2, k // 2, k // 2, k // 2, 0))
    ms = bench(fn)
    fl = 2.0 * B * H * H * Cin * Cout * k * k
    tot_t += ms * cnt; tot_f += fl * cnt
    print(f"conv{k}x{k} {H:3d}x{H:<3d} {Cin:5d}->{Cout:<5d} M={B*H*H:6d} K={Cin*k*k:6d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  x{cnt}")
lins = [  # (T, K, N, count)
    (4096, 320, 320, 25), (1024, 640, 640, 25), (256, 1280, 1280, 25), (64, 1280, 1280, 5), (77, 768, 320, 10), (77, 768, 640, 10), (77, 768, 1280, 12),
    (4096, 320, 2560, 5), (1024, 640, 5120, 5), (256, 1280, 10240, 5), (64, 1280, 10240, 1), (4096, 1280, 320, 5), (1024, 2560, 640, 5),
    (256, 5120, 1280, 5), (64, 5120, 1280, 1)]
for T, K, N, cnt in lins:
    a = g.to_dev((rng.standard_normal((B * T, K), dtype=np.float32)).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
    c = g.empty((B * T, N), f16)
    def fn():
        g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, B * T, N, K, 1, 0, 0, 0, 0))
    ms = bench(fn)
    fl = 2.0 * B * T * K * N
    tot_t += ms * cnt; tot_f += fl * cnt
    print(f"linear M={B*T:6d} K={K:5d} N={N:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  x{cnt}")
atts = [(8, 4096, 4096, 40, 5), (8, 1024, 1024, 80, 5), (8, 256, 256, 160, 5), (8, 64, 64, 160, 1), (8, 4096, 77, 40, 5), (8, 1024, 77, 80, 5), (8, 256, 77, 160, 5)]
for h, Tq, Tk, D, cnt in atts:
    q = g.to_dev(rng.standard_normal((B * h, Tq, D), dtype=np.float32).astype(f16))
    k_ = g.to_dev(rng.standard_normal((B * h, Tk, D), dtype=np.float32).astype(f16))
    v = g.to_dev(rng.standard_normal((B * h, Tk, D), dtype=np.float32).astype(f16))
    o = g.empty((B * h, Tq, D), f16)
    def fn():
        g._ck(g.lib.osg_attention(g.ctx, 2, q.ptr, k_.ptr, v.ptr, o.ptr, B * h, Tq, Tk, D, D ** -0.5, 0))
    ms = bench(fn)
    fl = 4.0 * B * h * Tq * Tk * D
    tot_t += ms * cnt; tot_f += fl * cnt
    print(f"attn h={B*h} Tq={Tq} Tk={Tk} D={D}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s  x{cnt}")
print(f"TOTAL contraction time {tot_t:.3f} ms for {tot_f/1e9:.1f} GFLOP -> {tot_f/tot_t/1e9:.1f} TF/s")
