"""CPU: the HOST logic of the backend -- model.txt parsing, fusion passes, lowering to the launch list, merged projections, arena packing,
opt-in plan variants (LayerNorm folding, side stream), error paths -- exercised without a GPU through a NO-OP stand-in for libosgpu.so
(tests/stub/make_stub.py: every C-ABI entry point exists, nothing is computed, outputs stay zero; reachable only through OSGPU_LIB, which
only this module sets).  What is checked is structure and invariants of the plan (Model.hip_plan_info), never numbers."""
import os
import sys
import tempfile

import numpy as np
import pytest

import golden_cases as gc
from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub"))


@pytest.fixture(scope="module")
def stub_backend():
    import make_stub
    from onnxstream_amd import build as b
    if not os.path.exists(b.LIB_HOST):
        pytest.skip("host library not built")
    with tempfile.TemporaryDirectory() as d:
        so = make_stub.build(d)
        old = os.environ.get("OSGPU_LIB")
        os.environ["OSGPU_LIB"] = so
        try:
            yield so
        finally:
            if old is None:
                os.environ.pop("OSGPU_LIB", None)
            else:
                os.environ["OSGPU_LIB"] = old


def _plan(model_dir, inputs, options=(), pushes=1):
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m.read_file(model_dir + "model.txt")
    for k, v in options:
        m._set_option(k, v)
    for _ in range(pushes):
        for k, v in inputs.items():
            m.add_tensor(k, v)
    m.set_use_fp16_arithmetic(True)
    m.set_fuse_ops_in_attention(True)
    m.run()
    info = m.hip_plan_info()
    m.run_count = 1
    return m, info


def _parse(info):
    steps, vals, arena = [], {}, 0
    for line in info.splitlines():
        if line.startswith("step "):
            head, what = line.split(" | ", 1)
            f = dict(kv.split("=") for kv in head.split()[2:])
            steps.append(dict(i=int(head.split()[1]), what=what,
                              reads=[int(v) for v in f["reads"].split(",") if v], writes=[int(v) for v in f["writes"].split(",") if v]))
        elif line.startswith("val "):
            f = dict(kv.split("=") for kv in line.split()[2:])
            vals[int(line.split()[1])] = {k: int(v) for k, v in f.items()}
        elif line.startswith("arena "):
            arena = int(line.split()[1])
    return steps, vals, arena


def _check_arena(steps, vals, arena):
    """no two activations that are alive at the same time share a byte; everything is 256-byte aligned and inside the arena"""
    items = sorted(vals.items(), key=lambda kv: kv[1]["offset"])
    for v, a in items:
        assert a["offset"] % 256 == 0 and a["offset"] + a["bytes"] <= arena, (v, a, arena)
        assert 0 <= a["first"] <= a["last"] < len(steps)
    for i, (v, a) in enumerate(items):
        for w, b in items[i + 1:]:
            if b["offset"] >= a["offset"] + ((a["bytes"] + 255) & ~255):
                break
            assert a["last"] < b["first"] or b["last"] < a["first"], ("live buffers overlap", v, a, w, b)
    # every activation a step touches is alive at that step
    for s in steps:
        for v in s["reads"] + s["writes"]:
            if v in vals:
                assert vals[v]["first"] <= s["i"] <= vals[v]["last"], (s, v, vals[v])


@pytest.mark.parametrize("name", gc.all_case_names())
def test_every_golden_graph_plans_at_every_fusion_level(stub_backend, name):
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    counts = {}
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        gc.emit(gc.by_name(name), DirSink(d))
        for tag, opts in (("f0", (("hip_fusion_level", 0),)), ("f1", (("hip_fusion_level", 1),)), ("f2", ()),
                          ("f2+ln", (("hip_fuse_ln_gemm", 1),))):
            m, info = _plan(d, ins, opts)
            steps, vals, arena = _parse(info)
            assert len(steps) == m.hip_last_kernel_count() and steps
            _check_arena(steps, vals, arena)
            counts[tag] = len(steps)
            out = m.get_tensor(str(z["out_name"]))
            assert out is not None and list(out[0].shape) == list(z["ref16"].shape)      # shape inference reached the graph output
            m.close()
    assert counts["f2"] <= counts["f1"] <= counts["f0"]
    assert counts["f2+ln"] <= counts["f2"]


def test_w8_resident_plan_keeps_the_fusions_of_the_f16_plan(stub_backend):
    """hip_w8_resident (uint8 weight codes resident, dequantised between the LDS tile and the MFMA; round 6: through the tuned kernels): the plan is the f16 plan of the
    same graph minus the two fusions that need f16 weights -- a LayerNorm folds gamma INTO the weight, the transformer-block tail streams f16 weights into registers --:
    merged projections concatenate codes and carry per-column (scale, zero point) vectors, the GEGLU rides in the epilogue on pair-interleaved codes, convolutions
    keep their output views; every contraction whose weight qualifies (K % 64 == 0) runs on codes."""
    name = "unet_tiny_w8"
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    plans = {}
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        gc.emit(gc.by_name(name), DirSink(d))
        for tag, opts in (("codes", (("hip_w8_resident", 1),)), ("f16", (("hip_fuse_ln_gemm", 0), ("hip_fuse_tblock", 0)))):
            m, info = _plan(d, ins, opts)
            steps, vals, arena = _parse(info)
            _check_arena(steps, vals, arena)
            plans[tag] = [s["what"] for s in steps]
            m.close()
    codes, f16p = plans["codes"], plans["f16"]
    assert len(codes) == len(f16p)
    kinds = lambda p: sorted(w.replace(" w8 ", " ").replace("Conv w8", "Conv").split(" ", 1)[0] for w in p)
    assert kinds(codes) == kinds(f16p)
    assert any(w.startswith("Linear w8 merged(") for w in codes) and any(w.startswith("Linear+GEGLU w8 ") for w in codes)
    assert sum(">concat" in w for w in codes) == sum(">concat" in w for w in f16p) > 0
    n_codes = sum(" w8 " in w or w.startswith("Conv w8") for w in codes)
    n_contr = sum(w.split(" ", 1)[0].split("+")[0] in ("Conv", "Linear", "Gemm") for w in codes)
    assert n_codes >= 0.7 * n_contr, (n_codes, n_contr)   # (the miniature's 32-channel level has K % 64 != 0 and stays f16, as conv_in's 4 input channels do everywhere)


def test_transformer_chain_plans(stub_backend):
    """The real-width transformer chains (tests/golden_cases.py CHAINS): at fusion level 2 the 320-wide one is proj_in, Q|K|V, self-attention and ONE osg_tblock_tail
    launch, which reads a K / V pack made by one KVPack launch; the 640- / 1280-wide ones keep the launches of round 3 (round 4's osg_qattn -- LayerNorm + attn2.to_q +
    cross-attention as one launch -- was removed in round 6: worth 0.01 ms of a pass).  With the tail fusion off the round-3 launches are back, and nothing else changes."""
    for name in ("transformer_block_320", "transformer_block_640", "transformer_block_1280"):
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
        ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
        with tempfile.TemporaryDirectory() as d:
            d += "/"
            gc.emit(gc.by_name(name), DirSink(d))
            m, info = _plan(d, ins, ())
            steps, vals, arena = _parse(info)
            m.close()
            _check_arena(steps, vals, arena)
            what = [s["what"] for s in steps]
            m, info = _plan(d, ins, (("hip_fuse_tblock", 0),))
            steps0 = _parse(info)[0]
            m.close()
            what0 = [s["what"] for s in steps0]
            assert not any(w.startswith(("TBlockTail", "KVPack")) for w in what0)
            assert sum(w.startswith("Attention ") for w in what0) == 2
            if name == "transformer_block_320":
                fused = [s for s in steps if s["what"].startswith("TBlockTail+proj_out ")]
                packs = [s for s in steps if s["what"].startswith("KVPack x1 ")]
                assert len(fused) == 1 and len(packs) == 1 and packs[0]["i"] < fused[0]["i"] and packs[0]["writes"][0] in fused[0]["reads"], what
                assert sum(w.startswith("Attention ") for w in what) == 1                   # the self-attention
                assert len(steps0) == len(steps) - 2 + 7, (name, len(steps0), len(steps))     # (- 2: the fused launch and the KVPack launch; + the 7 it replaces)
            else:
                assert what == what0


def test_unet_plan_structure(stub_backend):
    """The miniature UNet: what the fusion level 2 plan is made of, and what the opt-in variants change."""
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        m, info = _plan(d, ins, (("hip_fuse_ln_gemm", 0),), pushes=2)   # cond + uncond pushed under the same names = one batch-2 pass
        steps, vals, arena = _parse(info)
        kinds = [s["what"].split(" ", 1)[0].split("+")[0] for s in steps]
        m.close()
        assert kinds.count("Attention") == 2 * 16              # SD 1.5 topology: 16 transformer blocks, self + cross attention each, head split/merge fused in
        assert kinds.count("GroupNorm") == 61 and kinds.count("LayerNorm") == 3 * 16
        assert not any(k in ("Sigmoid", "Erf", "Softmax", "Transpose", "ReduceMean", "Pow", "InstanceNormalization") for k in kinds)
        assert sum("merged(" in s["what"] for s in steps) == 16 + 2     # Q|K|V per block, all cross K|V of the net, all time-embedding projections
        m2, info2 = _plan(d, ins, (("hip_fuse_ln_gemm", 1),), pushes=2)
        steps2, vals2, arena2 = _parse(info2)
        m2.close()
        n_ln = sum(s["what"].startswith("LayerNorm") for s in steps2)
        n_fold = sum(" ln+ " in s["what"] for s in steps2)
        assert n_fold > 0 and n_ln + n_fold == 3 * 16           # a LayerNorm either stays a launch (C % 64 != 0 at level 0) or rides in its consumer
        assert sum("+rowstats" in s["what"] for s in steps2) == n_fold      # and then its producer hands the row statistics over
        _check_arena(steps2, vals2, arena2)


def test_full_size_sd15_plan(stub_backend):
    """BASELINE's full-size graph (2 127 ops, 859.5 M parameters) through the planner: launch count, arena."""
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), "sd15") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_unet.build_unet(DirSink(d), sd_unet.SD15)
        open(d + ".complete", "w").write("ok")
    ins = sd_unet.unet_inputs(sd_unet.SD15, 42)
    m, info = _plan(d, ins, (("hip_fuse_ln_gemm", 0), ("hip_fuse_tblock", 0)), pushes=2)
    steps, vals, arena = _parse(info)
    m.close()
    assert len(steps) == 352                                    # (366 before round 3; the two time-embedding Gemm + SiLU pairs are one launch each; the 12 skip-connection Concats are no launches any more, see below)
    _check_arena(steps, vals, arena)
    assert arena < 400 * 2 ** 20                                # activations of a batch-2 pass pack into well under 400 MiB
    kinds = [s["what"].split(" ", 1)[0].split("+")[0] for s in steps]
    assert kinds.count("Attention") == 32 and kinds.count("GroupNorm") == 61 and kinds.count("LayerNorm") == 48
    m, info = _plan(d, ins, (("hip_fuse_tblock", 0),), pushes=2)   # round 3's default plan: every LayerNorm folded into its consuming GEMM
    steps_3 = _parse(info)[0]
    m.close()
    assert len(steps_3) == 304
    # round 4's default: at the 64 x 64 level (320 channels: the shape osg_tblock_tail takes) everything behind a block's self-attention -- to_out + residual,
    # LayerNorm, to_q, cross-attention, to_out + residual, LayerNorm, GEGLU projection, ff.net.2 + residual, proj_out + residual: 7 launches -- is ONE launch;
    # the K / V of all five blocks are re-packed by one launch right behind the merged K|V projection of the text context
    m, info = _plan(d, ins, (), pushes=2)
    steps_d, vals_d, arena_d = _parse(info)
    m.close()
    # ... the other levels (640 / 1280 channels, where a row block's weights are too many) keep their launches (round 4's osg_qattn was removed in round 6)
    assert len(steps_d) == 304 - 5 * 6 + 1
    tails = [s for s in steps_d if s["what"].startswith("TBlockTail+proj_out ")]
    packs = [s for s in steps_d if s["what"].startswith("KVPack x5 ")]
    assert len(tails) == 5 and len(packs) == 1 and all(s["i"] > packs[0]["i"] for s in tails)
    assert all(packs[0]["writes"][0] in s["reads"] for s in tails)     # every one of them reads the one pack buffer, which therefore lives until the last of them
    assert vals_d[packs[0]["writes"][0]]["last"] == max(s["i"] for s in tails)
    assert [s["what"].split(" ", 1)[0] for s in steps_d].count("Attention") == 32 - 5
    assert sum("attn2/to_q" in s["what"] for s in steps_d) == 11
    _check_arena(steps_d, vals_d, arena_d)
    # round 3: every skip-connection Concat of the up path is gone -- both of its operands come straight out of convolutions, which store into their
    # column slice of the concatenated buffer themselves (osg_conv2d_nhwc_v): 24 convolutions carry the mark, the only Concat launch left is the
    # [cos | sin] of the time embedding; with the option off the 12 copy launches are back
    kinds_d = [s["what"].split(" ", 1)[0] for s in steps_d]
    assert kinds_d.count("Concat") == 1 and sum(">concat" in s["what"] for s in steps_d) == 24
    for s in steps_d:
        if ">concat" in s["what"]:
            assert len(s["writes"]) in (1, 2)                   # the slice alone (the Concat was its only reader) or the dense tensor + the slice
    # a concatenated buffer is written by exactly two convolutions and lives from the earlier one (a down-path layer) to its last reader
    cat_vals = {}
    for s in steps_d:
        if ">concat" in s["what"]:
            cat_vals.setdefault(s["writes"][-1], []).append(s["i"])
    assert len(cat_vals) == 12 and all(len(v) == 2 for v in cat_vals.values())
    for v, writers in cat_vals.items():
        assert vals_d[v]["first"] == min(writers) and vals_d[v]["last"] > max(writers)
    m, info = _plan(d, ins, (("hip_concat_views", 0), ("hip_fuse_tblock", 0)), pushes=2)
    steps_o = _parse(info)[0]
    m.close()
    assert len(steps_o) == 316 and [s["what"].split(" ", 1)[0] for s in steps_o].count("Concat") == 13


def test_errors_are_the_reference_style_and_loud(stub_backend):
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model, OnnxStreamError
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        m.read_file(d + "model.txt")
        m._set_option("use_uint8_arithmetic", 1)
        for k, v in ins.items():
            m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        with pytest.raises(OnnxStreamError, match="uint8|range data|data type"):   # an fp16 UNet under uint8 arithmetic: no silent fallback
            m.run()
        m.close()
        # a graph with an op the backend does not implement
        txt = open(d + "model.txt").read().splitlines()
        bad = txt[0].replace(":Conv*", ":NonMaxSuppression*", 1) if ":Conv*" in txt[0] else None
        if bad:
            open(d + "model_bad.txt", "w").write("\n".join([bad] + txt[1:]) + "\n")
            os.replace(d + "model_bad.txt", d + "model.txt")
            m = Model(b.LIB_HOST, 0, "ram+nocache")
            m.read_file(d + "model.txt")
            for k, v in ins.items():
                m.add_tensor(k, v)
            m.set_use_fp16_arithmetic(True)
            with pytest.raises(OnnxStreamError, match="not implemented"):
                m.run()
            m.close()


@pytest.mark.parametrize("wp", ["ram+nocache", "nocache", "prefetch"])
def test_streamed_weights_mode_respects_the_provider_contract(stub_backend, wp):
    """hip_stream_weights: every pass pulls every weight through the WeightsProvider again, in strict model order, exactly once --
    DiskPrefetchWeightsProvider throws on any other sequence (reference src/onnxstream.h:570-573), so three clean passes ARE the check."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        m = Model(b.LIB_HOST, 0, wp)
        m.read_file(d + "model.txt")
        m._set_option("hip_stream_weights", 1)
        streamed = []
        for _ in range(3):
            for k, v in ins.items():
                m.add_tensor(k, v)
            m.set_use_fp16_arithmetic(True)
            m.set_fuse_ops_in_attention(True)
            m.run()
            streamed.append(m.hip_streamed_bytes())
            m.clear_tensors()
        m.close()
    assert streamed[0] == 0 and streamed[1] == streamed[2] > 0      # pass 1 makes the plan (resident upload), later passes re-stream everything


@pytest.mark.parametrize("ops_cache", [0, 1])
@pytest.mark.parametrize("wp", ["ram+nocache", "nocache", "prefetch", "ram+prefetch"])
def test_plan_rebuild_never_goes_back_to_the_provider(stub_backend, wp, ops_cache):
    """A plan is rebuilt when the number of pushed samples or a hip_* option changes (the reference allows the batch size to change
    between runs).  By then a strictly sequential provider is exhausted and, with m_use_ops_cache, the host copies were remove()d: the
    rebuild is served from the Model's pool of resident constants.  Also the round-1 advisor case: DiskPrefetch + use_ops_cache, where
    remove() shrinks the order DURING the first pass while the worker runs on its own snapshot."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        m = Model(b.LIB_HOST, 0, wp)
        m.read_file(d + "model.txt")
        m.set_use_ops_cache(bool(ops_cache))
        counts = []
        for pushes, opts in ((1, (("hip_fuse_ln_gemm", 0),)), (2, ()), (2, (("hip_fuse_ln_gemm", 1),)), (1, (("hip_fusion_level", 0),)), (3, (("hip_fusion_level", 2), ("hip_fuse_ln_gemm", 0)))):
            for k, v in opts:
                m._set_option(k, v)
            for _ in range(pushes):
                for k, v in ins.items():
                    m.add_tensor(k, v)
            m.set_use_fp16_arithmetic(True)
            m.set_fuse_ops_in_attention(True)
            m.run()
            counts.append(m.hip_last_kernel_count())
            out = m.get_tensor("out_sample", pushes - 1)
            assert out is not None
            m.clear_tensors()
        m.close()
    assert counts[0] == counts[1] == counts[4] and counts[2] < counts[1] < counts[3]


@pytest.mark.parametrize("wp", ["ram+nocache", "nocache", "prefetch"])
def test_vram_budget_streams_the_weights_beyond_it(stub_backend, wp):
    """CudaOptions::m_vram_to_use (reference src/onnxstream.cpp:396-398): weights stay resident in model order until the budget is spent,
    the rest are pulled from the provider and sent through the device ring EVERY pass, the first included; the provider's strict
    sequence survives (DiskPrefetch throws otherwise) and the device never holds more weight bytes than budget + ring."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        wbytes = sum(os.path.getsize(d + f) for f in os.listdir(d) if f.endswith(".bin"))
        budget = wbytes // 3
        m = Model(b.LIB_HOST, 0, wp)
        m.hip_set_vram_budget(budget)
        m.read_file(d + "model.txt")
        streamed = []
        for _ in range(3):
            for k, v in ins.items():
                m.add_tensor(k, v)
            m.set_use_fp16_arithmetic(True)
            m.set_fuse_ops_in_attention(True)
            m.run()
            streamed.append(m.hip_streamed_bytes())
            assert m.get_tensor("out_sample") is not None
            m.clear_tensors()
        held = m.hip_resident_weight_bytes()
        m.close()
    assert streamed[0] == streamed[1] == streamed[2] > 0.5 * wbytes           # everything beyond the budget, every pass
    assert streamed[0] < wbytes and held < wbytes                              # ... and the resident part is not re-sent / the footprint shrank


def test_device_sampler_loop_plumbing(stub_backend):
    """model_hip_sampler_loop / model_hip_set_input: argument checking and the 2-samples-per-prompt contract (no numbers: the stub computes nothing)."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model, OnnxStreamError
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    L = ins["sample"].shape
    steps = 3
    sc = [np.full(steps, v, np.float32) for v in (0.5, -2.0, 900.0, 2.0, -0.3, 0.1)]
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        m.read_file(d + "model.txt")
        x = np.zeros(L, np.float32)
        with pytest.raises(OnnxStreamError, match="no plan"):
            m.hip_sampler_loop("sample", "timestep", "out_sample", x, None, *sc)
        for _ in range(2):                                     # one prompt = cond + uncond pushed under the same names
            for k, v in ins.items():
                m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m.run()
        m.clear_tensors()
        noise = np.zeros((steps,) + L, np.float32)
        ms = m.hip_sampler_loop("sample", "timestep", "out_sample", x, noise, *sc)
        assert ms == 0.0 and np.isfinite(x).all()              # stub timers report 0; eps stays 0 => x stays 0
        m.hip_set_input("encoder_hidden_states", 1, ins["encoder_hidden_states"])
        with pytest.raises(OnnxStreamError, match="out of range"):
            m.hip_set_input("encoder_hidden_states", 2, ins["encoder_hidden_states"])
        with pytest.raises(OnnxStreamError, match="not found"):
            m.hip_set_input("no_such_input", 0, ins["sample"])
        with pytest.raises(OnnxStreamError, match="2 \\* prompts"):
            m.hip_sampler_loop("sample", "timestep", "out_sample", np.zeros((2,) + L[1:], np.float32), None, *sc)
        with pytest.raises(OnnxStreamError, match="not found"):
            m.hip_sampler_loop("sample", "timestep", "nope", x, None, *sc)
        m.close()


def test_input_contract_errors(stub_backend):
    """The caller contract of Model::run (reference src/onnxstream.cpp:3817-3842, :2618): missing input, inconsistent pushes, a shape that
    changed after the plan was built -- reference-style messages, no crash; and a plan survives clear_tensors / re-push."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model, OnnxStreamError
    ins = sd_unet.unet_inputs(sd_unet.TINY, 42)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), sd_unet.TINY)
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        m.read_file(d + "model.txt")
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m.add_tensor("sample", ins["sample"])
        m.add_tensor("timestep", ins["timestep"])
        with pytest.raises(OnnxStreamError, match="not found"):
            m.run()                                             # encoder_hidden_states was never pushed
        m.clear_tensors()
        for k, v in ins.items():
            m.add_tensor(k, v)
        m.add_tensor("sample", ins["sample"])                   # a second sample but no second context: inconsistent batch
        with pytest.raises(OnnxStreamError, match="inconsistent"):
            m.run()
        m.clear_tensors()
        for k, v in ins.items():
            m.add_tensor(k, v)
        m.run()
        n1 = m.hip_last_kernel_count()
        assert m.get_tensor("out_sample") is not None and m.get_tensor("no_such_tensor") is None
        m.clear_tensors()
        for k, v in ins.items():
            m.add_tensor(k, v)
        m.run()                                                 # same plan, second pass (this one captures the graph)
        assert m.hip_last_kernel_count() == n1
        m.clear_tensors()
        bad = dict(ins)
        bad["sample"] = np.zeros((1, 4, 8, 8), np.float32)
        for k, v in bad.items():
            m.add_tensor(k, v)
        with pytest.raises(OnnxStreamError):
            m.run()
        m.close()
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    with pytest.raises(OnnxStreamError):
        m.read_file("/nonexistent/model.txt")
    m.close()


_SWEEP = [
    sd_unet.UNetConfig(block_out=(32, 64), transformer_depth=(1, 1), heads=1, ctx_dim=16, ctx_len=7, latent=8, name="s0"),
    sd_unet.UNetConfig(block_out=(64, 64, 128), transformer_depth=(1, 0, 1), heads=4, ctx_dim=24, ctx_len=3, latent=16, name="s1"),
    sd_unet.UNetConfig(block_out=(32, 96, 96, 32), transformer_depth=(0, 1, 1, 0), heads=2, ctx_dim=40, ctx_len=77, latent=24, name="s2"),   # odd sizes: no halo kernel
    sd_unet.UNetConfig(block_out=(64, 128), transformer_depth=(2, 3), mid_depth=2, head_dim=32, ctx_dim=64, ctx_len=5, latent=8, linear_proj=True,
                       sdxl_add_embed=True, name="s3"),
    sd_unet.UNetConfig(block_out=(128,), transformer_depth=(1,), heads=8, ctx_dim=32, ctx_len=9, latent=32, layers_per_block=1, name="s4"),
    sd_unet.UNetConfig(block_out=(32, 32, 32, 32, 32), transformer_depth=(1, 1, 1, 1, 0), heads=2, ctx_dim=8, ctx_len=2, latent=32, name="s5"),
]


@pytest.mark.parametrize("cfg", _SWEEP, ids=[c.name for c in _SWEEP])
@pytest.mark.parametrize("pushes", [1, 3])
def test_planner_invariants_over_a_sweep_of_unet_shapes(stub_backend, cfg, pushes):
    """Structure fuzz: UNets of other depths / widths / head counts / odd spatial sizes, single and batched passes, every opt-in variant --
    the plan must build, keep the arena invariants, and stay consistent across variants."""
    ins = sd_unet.unet_inputs(cfg, 7)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_unet.build_unet(DirSink(d), cfg)
        base = None
        for opts in ((), (("hip_fuse_ln_gemm", 0),), (("hip_fusion_level", 0),)):
            m, info = _plan(d, ins, opts, pushes=pushes)
            steps, vals, arena = _parse(info)
            _check_arena(steps, vals, arena)
            out = m.get_tensor("out_sample")
            assert out is not None and list(out[0].shape) == [1, cfg.out_ch, cfg.latent, cfg.latent]
            if not opts:
                base = len(steps)
            elif opts[0][0] == "hip_fusion_level":
                assert len(steps) > base
            else:
                assert len(steps) >= base          # (standalone LayerNorm launches instead of the fold)
            m.close()


@pytest.mark.parametrize("level", [0, 2])
def test_full_size_uint8_vae_plan(stub_backend, level):
    """BASELINE config 3's W8A8 VAE decoder through the uint8 planner with the shipped range data: at fusion level >= 1 every GroupNorm block is
    ONE launch sequence (histogram, table, per-channel affine + SiLU with the table lookup inside); level 0 keeps one launch per graph op."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_vae
    cfg = sd_vae.SD_VAE
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), cfg.name + "_qu8") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_vae.build_vae_decoder(DirSink(d), cfg, quant_all=True)
        open(d + ".complete", "w").write("ok")
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m._set_option("hip_fusion_level", level)
    m.hip_read_range_data(os.path.join(os.path.dirname(os.path.abspath(b.__file__)), "synth", "data", cfg.name + "_qu8_range_data.txt"))
    m.set_use_uint8_arithmetic(True)
    m.read_file(d + "model.txt")
    m.add_tensor(cfg.in_name, sd_vae.vae_inputs(cfg)[cfg.in_name])
    m.run()
    steps, vals, arena = _parse(m.hip_plan_info())
    m.close()
    kinds = [s["what"].split(" qu8", 1)[0] for s in steps]
    if level == 0:
        assert kinds.count("NormAffineAct") == 0 and kinds.count("AffineAct") == 0 and kinds.count("InstanceNorm") == 30
    else:
        assert kinds.count("NormAffineAct") == 30 and kinds.count("AffineAct") == 0 and kinds.count("InstanceNorm") == 0
        # layouts: latents in, the attention block's [HW,C] detour, image out -- no transposes around the residual adds
        assert sum(k.startswith(("to_nchw", "to_nhwc")) for k in kinds) <= 4


@pytest.mark.parametrize("mode", ["f16", "u8"])
def test_a_second_call_with_the_same_inputs_does_not_plan_again(stub_backend, mode):
    """Round 3: the uint8 planner lowers its effective fusion level to 0 and compatible() compared THAT with the Model's request -- every uint8 call built a new
    plan (no captured pass, 1.8 ms of host work per decode on the GPU box).  One plan per Model, and from the second call on a captured pass to replay."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_vae
    cfg = sd_vae.TINY_VAE
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        sd_vae.build_vae_decoder(DirSink(d), cfg, quant_all=(mode == "u8"))
        z = sd_vae.vae_inputs(cfg)[cfg.in_name]
        if mode == "u8":     # (the ranges of the golden uint8 case: a calibration through the stub would measure nothing)
            g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vae_tiny_qu8.npz"))
            open(d + "range_data.txt", "w", newline="").write(str(g["ranges"]))
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        if mode == "u8":
            m.hip_read_range_data(d + "range_data.txt")
            m.set_use_uint8_arithmetic(True)
        else:
            m.set_use_fp16_arithmetic(True)
        m.read_file(d + "model.txt")
        for _ in range(3):
            m.add_tensor(cfg.in_name, z)
            m.run()
            m.clear_tensors()
        assert m.hip_plans_built() == 1
        m.add_tensor(cfg.in_name, z)
        m.hip_replay(1)                               # throws when no pass was captured
        m._set_option("hip_fusion_level", 1)          # a changed request does re-plan
        m.run()
        assert m.hip_plans_built() == 2
        m.close()


def test_group_norm_statistics_from_producers_plan(stub_backend):
    """hip_gn_stats = 1: in the full-size SD 1.5 plan the 31 GroupNorms of the 64 x 64 and 32 x 32 levels read what 33 convolutions add up in their epilogues
    (two of them through both destinations: the dense tensor for the next block, the Concat slot for the up path); with the default (large tensors only) and with 0 the plan has none."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    cfg = sd_unet.SD15
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), "sd15") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_unet.build_unet(DirSink(d), cfg)
        open(d + ".complete", "w").write("ok")
    for on in (0, 1, None):       # None: the Model's default (2 = tensors of >= 8 M elements only: none in this plan -- the largest GroupNorm input has 7.9 M)
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        if on is not None:
            m._set_option("hip_gn_stats", on)
        m._set_option("hip_fuse_tblock", 0)       # (a block tail fused into one launch cannot add statistics up: tested below)
        m.read_file(d + "model.txt")
        for i in (sd_unet.unet_inputs(cfg, 42), sd_unet.unet_inputs(cfg, 43)):
            for k, v in i.items():
                m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m.run()
        steps, vals, arena = _parse(m.hip_plan_info())
        m.close()
        what = [s["what"] for s in steps]
        assert len(steps) == 304
        assert sum(w.startswith("GroupNorm stats<") for w in what) == (31 if on else 0)
        assert sum(w.startswith("GroupNorm ") for w in what) == 61
        assert sum("+gnstats" in w for w in what) == (33 if on else 0)
        if on:      # every armed producer is a convolution that runs BEFORE the normalisation that reads its table
            first_gn = min(i for i, w in enumerate(what) if w.startswith("GroupNorm stats<"))
            assert any("+gnstats" in w for w in what[:first_gn])
            assert all(w.startswith("Conv ") for w in what if "+gnstats" in w)
    # with the block tails fused (the default), a GroupNorm whose input -- or one operand of the Concat it reads -- comes out of an osg_tblock_tail launch keeps
    # its own statistics pass (that launch's epilogue does not add statistics up): fewer armed producers, still convolutions only, same launch count
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m._set_option("hip_gn_stats", 1)
    m.read_file(d + "model.txt")
    for i in (sd_unet.unet_inputs(cfg, 42), sd_unet.unet_inputs(cfg, 43)):
        for k, v in i.items():
            m.add_tensor(k, v)
    m.set_use_fp16_arithmetic(True)
    m.set_fuse_ops_in_attention(True)
    m.run()
    what = [s["what"] for s in _parse(m.hip_plan_info())[0]]
    m.close()
    assert len(what) == 275 and 10 <= sum(w.startswith("GroupNorm stats<") for w in what) < 31   # (264 with round 4's osg_qattn, removed in round 6)
    assert all(w.startswith("Conv ") for w in what if "+gnstats" in w) and not any("+gnstats" in w for w in what if w.startswith("TBlockTail"))


def test_vae_decoder_plan_reads_group_norm_statistics_from_its_convolutions(stub_backend):
    """The Model's default (hip_gn_stats = 2) on the fp16 SD VAE decoder: its GroupNorms over 128 x 128 x 512 and larger tensors (>= 8 M elements: the throughput regime,
    where re-reading the tensor for the statistics is what the launch costs) take their statistics from the producing convolutions; the 64 x 64 x 512 ones do not."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_vae
    cfg = sd_vae.SD_VAE
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), cfg.name) + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_vae.build_vae_decoder(DirSink(d), cfg)
        open(d + ".complete", "w").write("ok")
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m.read_file(d + "model.txt")
    m.add_tensor(cfg.in_name, sd_vae.vae_inputs(cfg)[cfg.in_name])
    m.set_use_fp16_arithmetic(True)
    m.run()
    steps, vals, arena = _parse(m.hip_plan_info())
    m.close()
    what = [s["what"] for s in steps]
    fused = [w for w in what if w.startswith("GroupNorm stats<")]
    plain = [w for w in what if w.startswith("GroupNorm ") and not w.startswith("GroupNorm stats<")]
    assert len(fused) + len(plain) == 30 and len(fused) >= 18 and len(plain) >= 9
    assert all("mid_block" in w or "up_blocks.0" in w for w in plain)          # (the 64 x 64 level)
    assert sum("+gnstats" in w for w in what) >= len(fused)


