"""Dev tool (GPU box): error of one golden case under plan options.  usage: golden_err.py <case> [opt=val ...]"""
import os, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_cases as gc
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth.graph import DirSink
name = sys.argv[1]
z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
r16, r32, oname = z["ref16"], z["ref32"], str(z["out_name"])
for optset in [a for a in sys.argv[2:]] or [""]:
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        gc.emit(gc.by_name(name), DirSink(d))
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        m.read_file(d + "model.txt")
        for kv in filter(None, optset.split(",")):
            k, v = kv.split("=")
            m._set_option(k, int(v))
        for k, v in ins.items():
            m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
        m.run()
        got = m.get_tensor(oname)[0]
        m.close()
    mx = float(np.abs(r32).max())
    print(f"{name} [{optset}]: err16 {np.abs(got - r16).max() / mx:.3e}  err32 {np.abs(got - r32).max() / mx:.3e}  noise {np.abs(r16 - r32).max() / mx:.3e}")
