"""Dev tool (GPU box): one split-K GEMM of the UNet's low-resolution levels, back to back, by finish route: the reduce launch, the in-kernel fold with
its default bounded wait and with no wait at all (the last arriver folds the whole tile) over write-through slabs, and the XCD-local form of the same
(the slices of a tile on one XCD, slabs through its L2) with the default bound, none, and 50 us (a difference to the default means blocks DO run into
the bound)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu  # noqa: E402

os.environ["OSG_XCD_DEBUG"] = "1"
g = osgpu.Gpu(0)
rng = np.random.default_rng(0)
f16 = np.float16
REPS = 300
for M, K, N, cfg, s in [(512, 1280, 1280, 2, 4), (512, 5120, 1280, 2, 8), (512, 5120, 1280, 2, 4), (512, 1280, 3840, 2, 2), (128, 1280, 1280, 2, 8), (2048, 2560, 640, 2, 4),
                        (2048, 2560, 640, 1, 4), (512, 5120, 1280, 0, 8)]:
    a = g.to_dev((rng.standard_normal((M, K), dtype=np.float32)).astype(f16))
    w = g.to_dev((rng.standard_normal((N, K), dtype=np.float32) * 0.02).astype(f16))
    c = g.empty((M, N), f16)
    os.environ["OSG_GEMM_CFG"], os.environ["OSG_GEMM_SPLITS"] = str(cfg), str(s)
    row = []
    for name, tick, wait in [("s=1", None, None), ("reduce", "0", None), ("fold", "1", "500"), ("nowait", "1", "0"), ("xcd", "2", "500"), ("xcd-nowait", "2", "0"), ("xcd-50us", "2", "5000")]:
        if tick is None:
            os.environ["OSG_GEMM_SPLITS"] = "1"
        else:
            os.environ["OSG_GEMM_SPLITS"] = str(s)
            os.environ["OSG_SPLITK_TICKET"] = tick
        if wait is not None:
            os.environ["OSG_SPLITK_WAIT"] = wait

        def run():
            g._ck(g.lib.osg_gemm(g.ctx, 2, a.ptr, w.ptr, 1, None, 2, None, c.ptr, M, N, K, 1, 0, 0, 0, 0))
        for _ in range(20):
            run()
        g.sync()
        g.timer_start()
        for _ in range(REPS):
            run()
        us = g.timer_stop() * 1e3 / REPS
        row.append(f"{name} {us:7.2f}")
    print(f"M={M:5d} K={K:5d} N={N:5d} tile cfg {cfg} splits {s}:  " + "  ".join(row) + "  us", flush=True)
