"""Dev tool (GPU box): self-attention of the 64x64 level in the product's strided layout ([T, heads*D] projections)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
B = 2
for (T, h, D) in [(4096, 8, 40), (1024, 8, 80), (256, 8, 160)]:
    C = h * D
    qkv = g.to_dev(rng.standard_normal((B, T, 3 * C), dtype=np.float32).astype(f16))
    o = g.empty((B, T, C), f16)
    ld = 3 * C
    def fn():
        g._ck(g.lib.osg_attention_strided(g.ctx, 2, qkv.ptr, ld, D, T * ld, qkv.ptr + 2 * C, ld, D, T * ld, qkv.ptr + 4 * C, ld, D, T * ld,
                                          o.ptr, C, D, T * C, B, h, T, T, D, D ** -0.5))
    fn(); fn(); g.sync(); g.timer_start()
    for _ in range(20): fn()
    ms = g.timer_stop() / 20
    print(f"[V1={os.environ.get('OSG_ATTN_V1','0')}] attn T={T} h={B*h} D={D}: {ms*1e3:7.1f} us {4.0*B*h*T*T*D/ms/1e9:7.1f} TF/s")
