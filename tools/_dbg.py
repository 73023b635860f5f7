import faulthandler, sys, os, time
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0, os.getcwd())
import numpy as np
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink
cfg = sd_unet.TINY
d = "/tmp/synth_tiny/"
g, _ = sd_unet.build_unet(DirSink(d), cfg)
print("emitted", flush=True)
ins = sd_unet.unet_inputs(cfg, 42)
m = Model("onnxstream_amd/libonnxstream_amd.so", 0, "ram+nocache")
m.read_file(d + "model.txt")
for k, v in ins.items(): m.add_tensor(k, v)
m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
m._set_option("hip_fusion_level", int(sys.argv[1]))
m.set_ops_printf(True)
print("run", flush=True)
m.run()
print("done", flush=True)
o, s = m.get_tensor("out_sample"); print(s, np.abs(o).max())
