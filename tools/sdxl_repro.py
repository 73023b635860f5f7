"""Dev probe (GPU box): SDXL UNet, cond + uncond, four runs of one Model: md5 of every run's output (run 1 eager, 2.. replayed) -- the check that exposed the
missing LDS wait in front of the fused kernels' workgroup barriers in round 4 (profiles/r04_qattn_lds_barrier_race.txt; the kernel it showed up in, osg_qattn, was
removed in round 6)."""
import hashlib
import os
import sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink

d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), "sdxl") + "/"
if not os.path.exists(d + ".complete"):
    os.makedirs(d, exist_ok=True)
    sd_unet.build_unet(DirSink(d), sd_unet.SDXL)
    open(d + ".complete", "w").write("ok")
a, c = sd_unet.unet_inputs(sd_unet.SDXL, 42), sd_unet.unet_inputs(sd_unet.SDXL, 43)
for graph, fuse, lnfold in ((1, 1, 1),):
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m.read_file(d + "model.txt")
    m._set_option("hip_use_graph", graph)
    m._set_option("hip_fuse_ln_gemm", lnfold)
    sums = []
    first = None
    for r in range(4):
        for ins in (a, c):
            for k, v in ins.items():
                m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m.run()
        o = m.get_tensor("out_sample", 0)[0]
        if first is None:
            first = o
        sums.append(hashlib.md5(o.tobytes()).hexdigest()[:8] + f"({int((o != first).sum())} differ, max {float(np.abs(o - first).max()):.2e})")
        m.clear_tensors()
    m.close()
    print(f"hip_use_graph={graph} hip_fuse_ln_gemm={lnfold}: {sums}", flush=True)
