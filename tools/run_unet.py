"""Dev tool (GPU box): run the synthetic UNet on the HIP backend only. usage: run_unet.py CFG [batch] [runs] [fusion]"""
import ctypes
import os
import sys
import time


REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.bindings import Model  # noqa: E402
from onnxstream_amd.synth import sd_unet  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402

cfg = getattr(sd_unet, sys.argv[1] if len(sys.argv) > 1 else "TINY")
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
fusion = int(sys.argv[4]) if len(sys.argv) > 4 else 2
d = f"/tmp/synth_{cfg.name}/"
if not os.path.exists(d + "model.txt"):
    sd_unet.build_unet(DirSink(d), cfg)
ins = [sd_unet.unet_inputs(cfg, 42 + i) for i in range(batch)]
m = Model(os.path.join(REPO, "onnxstream_amd", "libonnxstream_amd.so"), 0, "ram+nocache")
m.read_file(d + "model.txt")
ms = m.lib.model_hip_last_pass_ms
ms.restype = ctypes.c_double
ms.argtypes = [ctypes.c_void_p]
for r in range(runs):
    for b in range(batch):
        for k, v in ins[b].items():
            m.add_tensor(k, v)
    if r == 0:
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m._set_option("hip_fusion_level", fusion)
    t0 = time.time()
    m.run()
    print(f"run {r}: wall {1e3 * (time.time() - t0):.2f} ms  device pass {ms(m.handle):.3f} ms", flush=True)
    m.clear_tensors()
m.close()
