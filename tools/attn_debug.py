"""Dev probe (GPU box): attn2_kernel error against the exact attention per head dim / key count (which stage of the kernel goes wrong: K = 0 makes P uniform,
V = 1 makes O = 1 whatever P is)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
from oracle import np_ops as ref
g = osgpu.Gpu(0)
f16 = np.float16
for D in [int(x) for x in os.environ.get("DS", "40,80,64").split(",")]:
    for Tkv in [64, 128, 256]:
        rng = np.random.default_rng(D + Tkv)
        heads, Tq = 1, 64
        q, k, v = [rng.standard_normal((heads, t, D), dtype=np.float32).astype(f16) for t in (Tq, Tkv, Tkv)]
        for what, kk, vv in (("K=0", k * 0, v), ("V=1", k, v * 0 + 1), ("Kd>=32=0", np.concatenate([k[..., :32], 0 * k[..., 32:]], -1), v),
                             ("Kd<32=0", np.concatenate([0 * k[..., :32], k[..., 32:]], -1), v), ("random", k, v)):
            want = ref.attention_exact(q, kk, vv, D ** -0.5).astype(np.float32)
            got = g.attention(g.to_dev(q), g.to_dev(kk), g.to_dev(vv), D ** -0.5, k_is_dt=False).numpy().astype(np.float32)
            err = np.abs(got - want) / np.abs(want).max()
            print(f"D={D} Tkv={Tkv} {what:9s}: max err {err.max():.3e}; err by 16-row block {[float(f'{err[0, i:i+16].max():.1e}') for i in range(0, Tq, 16)]}; by d block "
                  f"{[float(f'{err[..., i:i+8].max():.1e}') for i in range(0, D, 8)]}", flush=True)
