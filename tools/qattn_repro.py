"""Dev probe (GPU box): is osg_qattn reproducible on COLD operands?  Several (rows, weight) sets, a 1 GB fill between launches, every result compared with the
first result of its set."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnxstream_amd import osgpu
f16, f32 = np.float16, np.float32
gpu = osgpu.Gpu(0)
rng = np.random.default_rng(0)
rnd = lambda shape, std=1.0: (rng.standard_normal(shape, dtype=f32) * std).astype(f16)
big = gpu.empty((512 * 1024 * 1024,), f16)
for M, imgs, C, heads in ((8192, 2, 640, 10), (2048, 2, 640, 8), (2048, 2, 1280, 20), (512, 2, 1280, 8)):      # (run with OSG_QATTN_ANY_SIZE=1)
    Tk, nset = 77, 6
    sets = []
    for i in range(nset):
        x = gpu.to_dev(rnd((M, C), 1.5)); g = gpu.to_dev(rnd((C,))); b = gpu.to_dev(rnd((C,), 0.1))
        wk8 = gpu.tblock_pack_weight(gpu.to_dev(rnd((C, C), C ** -0.5)))
        k, v = gpu.to_dev(rnd((imgs, Tk, C))), gpu.to_dev(rnd((imgs, Tk, C)))
        kp, vtp = gpu.tblock_kv_pack(k, v, heads)
        sets.append((x, g, b, wk8, kp, vtp))
    scale = (C // heads) ** -0.5
    ref = []
    for s in sets:
        o, q = gpu.qattn(s[0], s[1], s[2], s[3], s[4], s[5], Tk, heads, scale, M // imgs, debug=True)
        ref.append((o.numpy(), q.numpy()))
    bad = 0
    for it in range(36):
        i = (it * 5 + 1) % nset
        gpu._ck(gpu.lib.osg_memset(gpu.ctx, big.ptr, it & 255, 1024 * 1024 * 1024))
        s = sets[i]
        o, q = gpu.qattn(s[0], s[1], s[2], s[3], s[4], s[5], Tk, heads, scale, M // imgs, debug=True)
        on, qn = o.numpy(), q.numpy()
        dq, do = int((qn != ref[i][1]).sum()), int((on != ref[i][0]).sum())
        if dq or do:
            bad += 1
            rows = np.unique(np.nonzero(qn != ref[i][1])[0])
            cols = np.unique(np.nonzero(qn != ref[i][1])[1])
            print(f"  launch {it} set {i}: q differs in {dq} elements (rows {rows[:8]}.. n={len(rows)}, cols {cols[:6]}.. n={len(cols)}), out in {do}")
    print(f"M={M} C={C} heads={heads}: {bad} of 36 cold launches differ from the first result of their set")
