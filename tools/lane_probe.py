"""Dev probe (GPU box): is a cond+uncond step faster as ONE batch-2 chain or as TWO concurrent batch-1 chains (two Models, two compute
streams, replayed from two host threads)?  Prints mean ms per step for: batch-2 alone, batch-1 alone, two batch-1 lanes concurrently.

    python tools/lane_probe.py [SD15|SDXL] [reps]
"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import sd_unet
from bench import ensure_model_dir

cfg = getattr(sd_unet, sys.argv[1] if len(sys.argv) > 1 else "SD15")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
autotune = int(os.environ.get("LANE_AUTOTUNE", "1"))
model_dir = ensure_model_dir(cfg, 0, lambda: None, False)
f32 = np.float32


def make(nsamples, seed0):
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m._set_option("hip_device", 0)
    m._set_option("hip_autotune", autotune)
    m.read_file(model_dir + "model.txt")
    m.set_use_fp16_arithmetic(True)
    m.set_fuse_ops_in_attention(True)
    for _ in range(2):
        for i in range(nsamples):
            for k, v in sd_unet.unet_inputs(cfg, seed0 + i).items():
                m.add_tensor(k, np.ascontiguousarray(v, f32))
        m.run()
        m.clear_tensors()
    return m


m2 = make(2, 42)
t2 = m2.hip_replay(reps)
print(f"batch-2 chain alone: {t2:.4f} ms/pass ({m2.hip_last_kernel_count()} launches)", flush=True)
a, c = make(1, 42), make(1, 43)
t1 = a.hip_replay(reps)
print(f"batch-1 chain alone: {t1:.4f} ms/pass ({a.hip_last_kernel_count()} launches)", flush=True)


def both(n):
    out = {}
    bar = threading.Barrier(3)

    def w(m, key):
        bar.wait()
        out[key] = m.hip_replay(n)
    th = [threading.Thread(target=w, args=(a, "a")), threading.Thread(target=w, args=(c, "c"))]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = (time.perf_counter() - t0) * 1e3 / n
    return wall, out


both(5)
for _ in range(3):
    wall, out = both(reps)
    print(f"two batch-1 lanes concurrently: wall {wall:.4f} ms per step (lane a {out['a']:.4f}, lane c {out['c']:.4f} ms/pass by their own events)", flush=True)
t2 = m2.hip_replay(reps)
print(f"batch-2 chain alone (again): {t2:.4f} ms/pass", flush=True)
