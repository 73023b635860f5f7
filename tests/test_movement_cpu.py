"""CPU: what the PLANNER hands the data-movement entry points, checked with real values.  The stub backend (tests/stub/make_stub.py) implements transpose,
strided copy, two-input concat, nearest resize, row gather and the f16 <-> f32 conversion in plain C (arithmetic launches still compute nothing), so a
graph of zero-FLOP ops -- Reshape, Transpose, Concat, Split, Slice, Resize, Gather, Unsqueeze / Squeeze / Flatten -- carries real numbers from the pushed
fp32 inputs to the fp32 outputs through the real lowering: shapes, permutations, pitches, offsets, batching.  Expected values: numpy on the f16-rounded
inputs (every op here moves f16 values without touching them; the same ops are compared with the reference's own output on the GPU, tests/test_golden.py)."""
import os
import sys
import tempfile

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "stub"))
from onnxstream_amd.synth.graph import DirSink, GraphBuilder  # noqa: E402

f32 = np.float32


@pytest.fixture(scope="module")
def stub_backend():
    import make_stub
    from onnxstream_amd import build as b
    if not os.path.exists(b.LIB_HOST):
        pytest.skip("host library not built")
    with tempfile.TemporaryDirectory() as d:
        old = os.environ.get("OSGPU_LIB")
        os.environ["OSGPU_LIB"] = make_stub.build(d)
        try:
            yield
        finally:
            if old is None:
                os.environ.pop("OSGPU_LIB", None)
            else:
                os.environ["OSGPU_LIB"] = old


def r16(x):
    return np.asarray(x, f32).astype(np.float16).astype(f32)


def _run(build, inputs, outputs, pushes=1, fusion=2):
    """build(g) emits the graph; inputs: name -> list of `pushes` fp32 arrays; returns name -> list of outputs (one per pushed sample)"""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        g = GraphBuilder(DirSink(d))
        build(g)
        g.finish()
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        m._set_option("hip_fusion_level", fusion)
        m.read_file(d + "model.txt")
        m.set_use_fp16_arithmetic(True)
        for k in range(pushes):
            for name, arrs in inputs.items():
                m.add_tensor(name, np.ascontiguousarray(arrs[k], f32))
        m.run()
        got = {}
        for o in outputs:      # (an output that depends on constants only has ONE sample however many were pushed)
            res = [m.get_tensor(o, i) for i in range(pushes)]
            got[o] = [r[0] for r in res if r is not None]
        m.close()
    return got


def _rn(seed, shape):
    return np.random.default_rng(seed).standard_normal(shape).astype(f32)


@pytest.mark.parametrize("pushes", [1, 3])
@pytest.mark.parametrize("fusion", [0, 2])
def test_reshape_transpose_concat_slice_chain(stub_backend, pushes, fusion):
    xs = [_rn(10 + k, (1, 6, 4, 10)) for k in range(pushes)]
    ys = [_rn(20 + k, (1, 6, 4, 3)) for k in range(pushes)]

    def build(g):
        x = g.input("x", (1, 6, 4, 10))
        y = g.input("y", (1, 6, 4, 3))
        c = g.concat("/cat_last", [x, y], 3)                         # [1,6,4,13]: innermost concat (one launch for two inputs)
        t = g.transpose("/t", c, (0, 2, 1, 3))                       # [1,4,6,13]
        r = g.reshape("/r", t, (1, 4, 78))
        s = g.slice_last("/s", r, 5, 70)                             # [1,4,65]
        t2 = g.transpose("/t2", s, (0, 2, 1))                        # [1,65,4]
        c2 = g.concat("/cat_mid", [t2, t2, t2], 1)                   # [1,195,4]: three inputs, middle axis
        g.op("/out", "Transpose", [c2], (4, 195, 1), {"perm": "2,1,0"}, out_names=["out"])
        g.op("/out_t", "Transpose", [t], (13, 6, 4, 1), {"perm": "3,2,1,0"}, out_names=["out_t"])

    got = _run(build, {"x": xs, "y": ys}, ["out", "out_t"], pushes, fusion)
    for k in range(pushes):
        c = np.concatenate([r16(xs[k]), r16(ys[k])], 3)
        t = c.transpose(0, 2, 1, 3)
        s = t.reshape(1, 4, 78)[:, :, 5:70]
        c2 = np.concatenate([s.transpose(0, 2, 1)] * 3, 1)
        assert np.array_equal(got["out"][k], c2.transpose(2, 1, 0))
        assert np.array_equal(got["out_t"][k], t.transpose(3, 2, 1, 0))


@pytest.mark.parametrize("pushes", [1, 2])
def test_split_resize_gather_unsqueeze(stub_backend, pushes):
    xs = [_rn(30 + k, (1, 8, 5, 7)) for k in range(pushes)]
    table = _rn(3, (11, 6))
    ids = np.asarray([[4, 0, 10, 10, 7]], np.int64)

    def build(g):
        x = g.input("x", (1, 8, 5, 7))
        a, b_, c = g.op("/split", "Split", [x, g.const_i64("/split.sizes", [3, 1, 4])], [(1, 3, 5, 7), (1, 1, 5, 7), (1, 4, 5, 7)], {"axis": "1"})
        up = g.op("/up", "Resize", [c, None, g.weight("/up.scales", np.asarray([1, 1, 2, 2], f32), dtype="float32")], (1, 4, 10, 14),
                  {"coordinate_transformation_mode": "asymmetric", "mode": "nearest", "nearest_mode": "floor"})
        g.op("/up_out", "Transpose", [up], (1, 10, 14, 4), {"perm": "0,2,3,1"}, out_names=["up"])
        sq = g.op("/sq", "Squeeze", [b_, g.const_i64("/sq.axes", [1])], (1, 5, 7))
        g.op("/a_out", "Transpose", [g.unsqueeze("/unsq", a, 0)], (1, 7, 5, 3, 1), {"perm": "0,4,3,2,1"}, out_names=["a"])
        g.op("/b_out", "Transpose", [sq], (7, 5, 1), {"perm": "2,1,0"}, out_names=["b"])
        e = g.op("/emb", "Gather", [g.weight("/emb.weight", table), g.weight("/emb.idx", ids, dtype="int64")], (1, 5, 6), {"axis": "0"})
        g.op("/e_out", "Transpose", [e], (6, 5, 1), {"perm": "2,1,0"}, out_names=["e"])

    got = _run(build, {"x": xs}, ["up", "a", "b", "e"], pushes)
    for k in range(pushes):
        x = r16(xs[k])
        a, b_, c = x[:, :3], x[:, 3:4], x[:, 4:]
        up = c.repeat(2, axis=2).repeat(2, axis=3)
        assert np.array_equal(got["up"][k], up.transpose(0, 2, 3, 1))
        assert np.array_equal(got["a"][k], a[None].transpose(0, 4, 3, 2, 1))
        assert np.array_equal(got["b"][k], b_[:, 0].transpose(2, 1, 0))
    assert len(got["e"]) == 1 and np.array_equal(got["e"][0], r16(table)[ids[0]][None].transpose(2, 1, 0))


@pytest.mark.parametrize("pushes", [1, 2])
def test_layout_aliases_and_their_materialisation(stub_backend, pushes):
    """Transpose(0,3,1,2) of a plain [n,H,W,C] tensor is an ALIAS in the channels-last layout; a Reshape behind it needs the logical image and makes the
    planner materialise it (Plan::ensure_plain -> osg_transpose per sample); Transpose(0,2,3,1) of a channels-last value is an alias back"""
    xs = [_rn(40 + k, (1, 5, 6, 8)) for k in range(pushes)]

    def build(g):
        x = g.input("x", (1, 5, 6, 8))
        t = g.transpose("/to_cf", x, (0, 3, 1, 2))                   # [1,8,5,6], stored as it was
        r = g.reshape("/r", t, (1, 8, 30))
        g.op("/r_out", "Transpose", [r], (30, 8, 1), {"perm": "2,1,0"}, out_names=["r"])
        back = g.transpose("/to_cl", t, (0, 2, 3, 1))                # [1,5,6,8] again
        g.op("/back_out", "Transpose", [back], (1, 6, 5, 8), {"perm": "0,2,1,3"}, out_names=["back"])

    got = _run(build, {"x": xs}, ["r", "back"], pushes)
    for k in range(pushes):
        x = r16(xs[k])
        assert np.array_equal(got["r"][k], x.transpose(0, 3, 1, 2).reshape(1, 8, 30).transpose(2, 1, 0))
        assert np.array_equal(got["back"][k], x.transpose(0, 2, 1, 3))
