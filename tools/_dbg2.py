import faulthandler, sys, os, time
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.getcwd())
import numpy as np
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink
cfg = sd_unet.TINY
d = "/tmp/synth_tiny/"
if not os.path.exists(d + "model.txt"):
    sd_unet.build_unet(DirSink(d), cfg)
ins = sd_unet.unet_inputs(cfg, 42)
which = sys.argv[1]
if which == "ref":
    print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
    try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
    except Exception as e: print(e)
    os.system("nproc; lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket' ; cat /proc/loadavg")
    m = Model("oracle/_ref/libonnxstream_ref.so", int(sys.argv[2]), "ram+nocache")
else:
    m = Model("onnxstream_amd/libonnxstream_amd.so", 0, "ram+nocache")
m.read_file(d + "model.txt")
for r in range(4):
    for k, v in ins.items(): m.add_tensor(k, v)
    if r == 0:
        m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
    print("run", r, flush=True)
    t0 = time.time(); m.run(); print("done", r, time.time() - t0, flush=True)
    o, s = m.get_tensor("out_sample"); print(s, np.abs(o).max(), flush=True)
    m.clear_tensors()
