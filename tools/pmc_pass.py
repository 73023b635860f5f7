"""Dev tool (GPU box): a few EAGER (uncaptured, autotune off unless OSG_TUNE_CACHE is primed) SD 1.5 UNet batch-2 passes, the process
rocprofv3 --pmc wraps (tools/pmc_round.sh): every kernel of a pass is an ordinary dispatch the counters can attribute."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink
cfg = getattr(sd_unet, os.environ.get("PMC_CONFIG", "SD15"))
w8 = os.environ.get("PMC_W8") == "1"      # W8A16 with the weight codes resident (uint8 weights in model.txt, hip_w8_resident)
d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), cfg.name + ("_w8" if w8 else "")) + "/"
if not os.path.exists(d + ".complete"):
    os.makedirs(d, exist_ok=True)
    sd_unet.build_unet(DirSink(d), cfg, quant_weights=w8)
    open(d + ".complete", "w").write("ok")
passes = int(os.environ.get("PMC_PASSES", "2"))
m = Model(b.LIB_HOST, 0, "ram+nocache")
m.read_file(d + "model.txt")
m._set_option("hip_use_graph", 0)
if w8:
    m._set_option("hip_w8_resident", 1)
m._set_option("hip_autotune", int(os.environ.get("PMC_AUTOTUNE", "0")))
for r in range(passes):
    for s in (42, 43):
        for k, v in sd_unet.unet_inputs(cfg, s).items():
            m.add_tensor(k, v)
    m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
    m.run()
    m.clear_tensors()
print("passes", passes, "launches", m.hip_last_kernel_count(), "last pass ms", m.hip_last_pass_ms())
m.close()
