"""Generate tests/golden/sdpa_*.npz from the REFERENCE (oracle/_ref): the ScaledDotProductAttention chains of tests/sdpa_cases.py run
UNFUSED (m_use_scaled_dp_attn_op off -- the oracle's XNNPACK has no SDPA operator) with fp16 and with fp32 arithmetic."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import sdpa_cases as sc  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402
from oracle import ref as oref  # noqa: E402

assert oref.available(), "build the oracle first: make -C oracle ref"
for case in sc.CASES:
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        ins = sc.emit(case, DirSink(d))
        o16 = oref.run_model(d, ins, fp16=True, fuse_attention=False, threads=1)
        o32 = oref.run_model(d, ins, fp16=False, fuse_attention=False, threads=1)
        (oname, v16), = o16.items()
        v32 = o32[oname]
        mx = float(np.abs(v32).max())
        print(f"{case.__name__:14s} out={oname} shape={v16.shape} max|ref32|={mx:.3f} |ref16-ref32|/max={np.abs(v16 - v32).max() / mx:.2e}")
        np.savez_compressed(os.path.join(REPO, "tests", "golden", case.__name__ + ".npz"), out_name=np.asarray(oname), ref16=v16, ref32=v32,
                            **{"in_" + k: v for k, v in ins.items()})
