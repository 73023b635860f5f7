"""Dev probe (GPU box): osg_qattn (LayerNorm + attn2.to_q + cross-attention in one launch) against the same three ops as separate launches, at the SD 1.5
32x32 / 16x16 / 8x8 levels, cycling through enough weight sets that every launch finds its weights cold (as inside a pass)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnxstream_amd import osgpu

f16, f32 = np.float16, np.float32


def main():
    gpu = osgpu.Gpu(0)
    gpu.lib.osg_set_autotune(gpu.ctx, 1)
    rng = np.random.default_rng(0)
    rnd = lambda shape, std=1.0: (rng.standard_normal(shape, dtype=f32) * std).astype(f16)
    heads, imgs, Tk = 8, 2, 77
    for M, C in ((2048, 640), (512, 1280), (128, 1280)):
        nsets = 48 if C == 640 else 24
        x = gpu.to_dev(rnd((M, C)))
        g, b = gpu.to_dev(rnd((C,))), gpu.to_dev(rnd((C,), 0.1))
        w0 = rnd((C, C), C ** -0.5)
        wnk = [gpu.to_dev(np.roll(w0, i, axis=0)) for i in range(nsets)]
        wk8 = [gpu.tblock_pack_weight(w) for w in wnk]
        k, v = gpu.to_dev(rnd((imgs, Tk, C))), gpu.to_dev(rnd((imgs, Tk, C)))
        kp, vtp = gpu.tblock_kv_pack(k, v, heads)
        scale = (C // heads) ** -0.5
        out = gpu.empty((M, C), f16)
        for name, ns in (("cold", nsets), ("hot", 1)):
            for _ in range(3):
                gpu.qattn(x, g, b, wk8[0], kp, vtp, Tk, heads, scale, M // imgs, out=out)
            gpu.sync(); gpu.timer_start()
            n = 0
            for r in range(4):
                for i in range(ns if ns > 1 else 24):
                    gpu.qattn(x, g, b, wk8[i % ns], kp, vtp, Tk, heads, scale, M // imgs, out=out)
                    n += 1
            t_f = 1000 * gpu.timer_stop() / n
            for _ in range(3):
                n2 = gpu.layer_norm(x, g, b, 1e-5)
                q = gpu.gemm(n2, wnk[0], None, None, b_is_nk=True)
            gpu.sync(); gpu.timer_start()
            n = 0
            for r in range(4):
                for i in range(ns if ns > 1 else 24):
                    n2 = gpu.layer_norm(x, g, b, 1e-5)
                    q = gpu.gemm(n2, wnk[i % ns], None, None, b_is_nk=True)
                    q.shape = (imgs, M // imgs, C)
                    a2 = gpu.attention_tokens(q, k, v, heads, scale)
                    n += 1
            t_s = 1000 * gpu.timer_stop() / n
            print(f"M={M} C={C} {name}: osg_qattn {t_f:6.1f} us | LayerNorm + GEMM + attention launches {t_s:6.1f} us (host-paced: includes their output allocations)")


if __name__ == "__main__":
    main()
