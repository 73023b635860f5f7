"""Generate tests/golden/maskops.npz from the REFERENCE (oracle/_ref): tests/maskops_case.py in fp16 and fp32 arithmetic, lengths 6, 9, 6."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import maskops_case as mc  # noqa: E402
from onnxstream_amd.synth.graph import DirSink, GraphBuilder  # noqa: E402
from oracle import ref as oref  # noqa: E402

assert oref.available()
with tempfile.TemporaryDirectory() as d:
    d += "/"
    mc.build(GraphBuilder(DirSink(d), seed=1))
    o16, _ = mc.run(oref.REF_LIB, d, True)
    o32, _ = mc.run(oref.REF_LIB, d, False)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "maskops.npz"), **{f"ref16_{i}": a for i, a in enumerate(o16)}, **{f"ref32_{i}": a for i, a in enumerate(o32)})
print([a.shape for a in o16]); print(o16[0][0, 0])
