"""A graph made of the plan-time ops the LLM flow does NOT exercise (ConstantOfShape, Trilu, Equal, Greater, And, Expand on int64, Neg on a
plan-time fp32 tensor, nested Unsqueeze) next to the ones it does (Shape, Gather, Concat, Range, Where), within the forms the reference
implements (src/onnxstream.cpp :7543, :7883, :7637, :7154, :7034): a [T,T] additive mask built from the LENGTH of an int64 input and added to
a float input.  Fixture: the reference's own output (tools/make_golden_maskops.py -> tests/golden/maskops.npz)."""
from __future__ import annotations

import numpy as np

from onnxstream_amd.synth.graph import GraphBuilder

T = 6


def build(g: GraphBuilder):
    ids = g.input("ids", (1, 0))
    x = g.input("x", (1, 1, 0, 0))
    i64 = g.const_i64

    def sc(n, v):
        return g.weight(n, np.asarray(v, dtype=np.int64).reshape(()), dtype="int64")
    sh = g.op("/Shape", "Shape", [ids], (2,))
    t = g.op("/Gather", "Gather", [sh, sc("one", 1)], [()], {"axis": "0"})[0]
    t1 = g.op("/Unsqueeze", "Unsqueeze", [t, i64("ax0", [0])], (1,))
    shp = g.op("/Concat", "Concat", [t1, t1], (2,), {"axis": "0"})
    full = g.op("/ConstantOfShape", "ConstantOfShape", [shp], (0, 0), {"value": "-65504"})
    tri = g.op("/Trilu", "Trilu", [full, sc("k1", 1)], (0, 0), {"upper": "1"})
    r = g.op("/Range", "Range", [sc("zero", 0), t, sc("rone", 1)], (0,))
    rcol = g.op("/Unsqueeze_1", "Unsqueeze", [r, i64("ax1", [1])], (0, 1))
    eq = g.op("/Equal", "Equal", [r, rcol], (0, 0))
    gt = g.op("/Greater", "Greater", [r, rcol], (0, 0))
    ex = g.op("/Expand", "Expand", [rcol, shp], (0, 0))
    ge = g.op("/Greater_1", "Greater", [ex, gt], (0, 0))
    both = g.op("/And", "And", [ge, eq], (0, 0))
    w = g.op("/Where", "Where", [both, g.scalar("two", 2.0), g.scalar("mone", -1.0)], (0, 0))
    nw = g.op("/Neg", "Neg", [w], (0, 0))
    s = g.op("/Add", "Add", [tri, nw], (0, 0))
    s3 = g.op("/Unsqueeze_2", "Unsqueeze", [s, i64("ax00", [0])], (1, 0, 0))
    s4 = g.op("/Unsqueeze_3", "Unsqueeze", [s3, i64("ax000", [0])], (1, 1, 0, 0))
    g.op("/out", "Add", [x, s4], (1, 1, 0, 0), out_names=["out"])
    g.finish()


def inputs(t=T):
    return {"ids": np.zeros((1, t), np.int64), "x": np.random.default_rng(3).standard_normal((1, 1, t, t)).astype(np.float32)}


def run(lib, model_dir, fp16=True, t=T, options=()):
    from onnxstream_amd.bindings import Model
    m = Model(lib, 1, "ram+nocache")
    for k, v in options:
        m._set_option(k, v)
    m.set_support_dynamic_shapes(True)
    m.read_file(model_dir + "model.txt")
    outs = []
    for tt in (t, t + 3, t):                       # three calls with two lengths: the second and third re-plan
        m.set_use_fp16_arithmetic(False)
        for k, v in inputs(tt).items():
            m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(fp16)
        m.run()
        outs.append(m.get_tensor("out")[0])
        m.clear_tensors()
    info = m.hip_plan_info() if hasattr(m._lib, "model_hip_plan_info") else ""
    m.close()
    return outs, info
