"""Plan-time evaluation of the mask / shape op family against the reference (tests/maskops_case.py): ConstantOfShape, Trilu, Equal, Greater,
And, Expand, Where, Neg, Range, Shape / Gather / Concat / Unsqueeze -- everything folds while the plan is built, one device Add remains."""
import os
import sys
import tempfile

import numpy as np
import pytest

import maskops_case as mc
from onnxstream_amd.synth.graph import DirSink, GraphBuilder

HERE = os.path.dirname(os.path.abspath(__file__))
Z = np.load(os.path.join(HERE, "golden", "maskops.npz"))


def test_reference_reproduces_maskops_golden():
    from oracle import ref as oref
    if not oref.available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        mc.build(GraphBuilder(DirSink(d), seed=1))
        o16, _ = mc.run(oref.REF_LIB, d, True)
    for i, a in enumerate(o16):
        assert np.array_equal(a, Z[f"ref16_{i}"])


sys.path.insert(0, os.path.join(HERE, "stub"))


def test_maskops_fold_at_plan_time():
    import make_stub
    from onnxstream_amd import build as b
    if not os.path.exists(b.LIB_HOST):
        pytest.skip("host library not built")
    with tempfile.TemporaryDirectory() as d:
        old = os.environ.get("OSGPU_LIB")
        os.environ["OSGPU_LIB"] = make_stub.build(d)
        try:
            d += "/"
            mc.build(GraphBuilder(DirSink(d), seed=1))
            outs, info = mc.run(b.LIB_HOST, d, True)
        finally:
            if old is None:
                os.environ.pop("OSGPU_LIB", None)
            else:
                os.environ["OSGPU_LIB"] = old
    whats = [ln.split(" | ", 1)[1] for ln in info.splitlines() if ln.startswith("step ")]
    assert [o.shape for o in outs] == [(1, 1, 6, 6), (1, 1, 9, 9), (1, 1, 6, 6)]
    assert [w.split(" ")[0] for w in whats] == ["input", "Add", "output"], whats


@pytest.mark.gpu
@pytest.mark.parametrize("fusion", [0, 2])
def test_hip_maskops_equal_the_reference_bit_for_bit(fusion):
    """the folded mask holds the reference's values exactly (0, +-1, -2, -65504 and their sums are f16 numbers) and the one remaining op is a single
    f16 Add: the output is the reference's fp16 output to the bit, at both lengths, and again after re-planning back"""
    from onnxstream_amd import build as b
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        mc.build(GraphBuilder(DirSink(d), seed=1))
        outs, _ = mc.run(b.LIB_HOST, d, True, options=(("hip_fusion_level", fusion),))
    for i, a in enumerate(outs):
        assert np.array_equal(a, Z[f"ref16_{i}"]), (i, float(np.abs(a - Z[f"ref16_{i}"]).max()))
