"""Dev tool (GPU box): error table of every golden case under deterministic plans (hip_autotune=0), and -- where oracle/_ref travelled --
the reference's own fp16 output on THIS host (XNNPACK picks kernels per CPU: the oracle moves between hosts), saved for the fixtures.
usage: golden_table.py [out.npz]"""
import os, sys, tempfile
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_cases as gc
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth.graph import DirSink
from oracle import ref as oref

alt = {}
for name in gc.all_case_names():
    z = np.load(os.path.join(REPO, "tests", "golden", name + ".npz"))
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    r16, r32, oname = z["ref16"], z["ref32"], str(z["out_name"])
    mx = float(np.abs(r32).max())
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        gc.emit(gc.by_name(name), DirSink(d))
        line = f"{name:18s} noise {np.abs(r16 - r32).max() / mx:.3e}"
        if oref.available():
            h16 = oref.run_model(d, ins, fp16=True)[oname]
            h32 = oref.run_model(d, ins, fp16=False)[oname]
            alt[name] = h16
            line += f" | host ref16 vs fixture {np.abs(h16 - r16).max() / mx:.3e} host noise {np.abs(h16 - r32).max() / mx:.3e} ref32 same {np.array_equal(h32, r32)}"
        for fusion in (0, 1, 2):
            for extra in ({}, {"hip_fuse_ln_gemm": 0}) if fusion == 2 else ({},):
                m = Model(b.LIB_HOST, 0, "ram+nocache")
                m.read_file(d + "model.txt")
                m._set_option("hip_fusion_level", fusion)
                m._set_option("hip_autotune", 0)
                for k, v in extra.items():
                    m._set_option(k, v)
                for k, v in ins.items():
                    m.add_tensor(k, v)
                m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
                m.run()
                got = m.get_tensor(oname)[0]
                m.close()
                e16 = np.abs(got - r16).max() / mx
                e32 = np.abs(got - r32).max() / mx
                ea = np.abs(got - alt[name]).max() / mx if name in alt else float("nan")
                line += f" | f{fusion}{'-lnfold' if extra else ''}: e16 {e16:.2e} e16host {ea:.2e} e32 {e32:.2e}"
    print(line, flush=True)
if alt and len(sys.argv) > 1:
    np.savez_compressed(sys.argv[1], **alt)
