"""-m gpu, BASELINE.json's full size: the complete SD 1.5 UNet (2 127 graph ops, 859.5 M parameters, 2x4x64x64 latents).

* against the reference itself (oracle/_ref travels to the GPU box; one fp16 and one fp32 CPU pass, a few seconds each) AND the committed fp16 outputs of
  the reference on two hosts (tests/golden/ref16_fullsize_{xeon,epyc}.npz, tools/ref16_fullsize.py): the rule of tests/parity.py at full size, whose third
  leg is the PINNED host-to-host spread of the reference itself;
* size-independent properties: bitwise reproducibility over eager / captured / replayed passes, and batch invariance --
  a sample's result does not depend on what else shares the batched pass (cond alone == cond next to uncond)."""
import os

import numpy as np
import pytest

from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink
from oracle import ref as oref
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd15_dir():
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), "sd15") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_unet.build_unet(DirSink(d), sd_unet.SD15)
        open(d + ".complete", "w").write("ok")
    return d


def _run(lib, d, pushes, runs=1, options=()):
    from onnxstream_amd.bindings import Model
    m = Model(lib, 0, "ram+nocache")
    m.read_file(d + "model.txt")
    for k, v in options:
        m._set_option(k, v)
    outs = []
    for r in range(runs):
        for ins in pushes:
            for k, v in ins.items():
                m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m.run()
        outs.append([m.get_tensor("out_sample", i)[0] for i in range(len(pushes))])
        m.clear_tensors()
    m.close()
    return outs


def test_sd15_unet_properties_and_reference_parity(sd15_dir):
    from onnxstream_amd import build as b
    a, c = sd_unet.unet_inputs(sd_unet.SD15, 42), sd_unet.unet_inputs(sd_unet.SD15, 43)
    both = _run(b.LIB_HOST, sd15_dir, [a, c], runs=3)
    for o in both[1:]:
        assert np.array_equal(both[0][0], o[0]) and np.array_equal(both[0][1], o[1])      # eager == captured == replayed, bit for bit
    alone = _run(b.LIB_HOST, sd15_dir, [a])[0][0]
    mx = float(np.abs(alone).max())
    # batch invariance: tile shapes / split-K choices may differ between M = 4096 and M = 8192 launches, roundings may not by more than f16 noise
    assert float(np.abs(alone - both[0][0]).max()) / mx <= 5e-3
    assert np.isfinite(both[0][0]).all() and np.isfinite(both[0][1]).all()
    # opt-in LayerNorm folding (48 LayerNorms -> epilogues of their consuming GEMMs, row statistics handed over by the producing GEMMs):
    # same result up to the f16 rounding of the normalised activation it no longer materialises
    folded = _run(b.LIB_HOST, sd15_dir, [a, c], options=(("hip_fuse_ln_gemm", 1),))[0]
    assert float(np.abs(folded[0] - both[0][0]).max()) / mx <= 5e-3 and float(np.abs(folded[1] - both[0][1]).max()) / float(np.abs(both[0][1]).max()) <= 5e-3
    # opt-in GroupNorm statistics from the producing convolutions' epilogues (31 normalisations read int64 tables that 33 epilogues / split-K reduce launches
    # fill with integer atomics): the statistics are sums of the same f16 values in another order, so f16 noise apart the same result -- and, the additions
    # being integer, the same BITS eager, captured and replayed
    gns = _run(b.LIB_HOST, sd15_dir, [a, c], runs=3, options=(("hip_gn_stats", 1),))
    for o in gns[1:]:
        assert np.array_equal(gns[0][0], o[0]) and np.array_equal(gns[0][1], o[1])
    assert float(np.abs(gns[0][0] - both[0][0]).max()) / mx <= 5e-3 and float(np.abs(gns[0][1] - both[0][1]).max()) / float(np.abs(both[0][1]).max()) <= 5e-3
    if not oref.available():
        pytest.skip("oracle/_ref not present: properties checked, reference parity skipped")
    _triangulated(both[0][0], sd15_dir, a, "SD1.5 UNet full size", "sd15")


def _triangulated(got, d, ins, what, case, out="out_sample", sub=None):
    """The rule of tests/parity.py at full size: the reference's fp16 and fp32 passes run HERE (this host's XNNPACK micro-kernels), next to the committed fp16
    outputs of the hosts of tests/golden/ref16_fullsize_*.npz (used only when their fp32 output equals this host's bit for bit: same model, same input)."""
    r16 = oref.run_model(d, ins, fp16=True)[out]
    r32 = oref.run_model(d, ins, fp16=False)[out]
    refs = [r16]
    fix = parity.fullsize_host_refs(case, r32, sub)
    if sub is None:
        refs += [f for f in fix if not np.array_equal(f, r16)]
        err16, err32, drift, spread = parity.triangulate(got, refs, r32)
    else:   # the fixture holds a SUBSAMPLE of the output: err16 / drift against this host on the whole tensor; the spread between the hosts on the subsample
            # (a lower bound of the whole tensor's: conservative as a bound)
        err16, err32, drift, _ = parity.triangulate(got, refs, r32)
        hosts = [r16[sub]] + [f for f in fix if not np.array_equal(f, r16[sub])]
        spread = None
        if len(hosts) >= 2:
            spread = max(float(np.abs(hosts[i] - hosts[j]).max()) for i in range(len(hosts)) for j in range(i + 1, len(hosts))) / float(np.abs(r32).max())
    print(f"{what}: {len(refs) if sub is None else len(hosts)} host(s) of the reference")
    parity.check(what, err16, err32, drift, key=what, spread=spread)


def test_sd15_unet_tuned_plan_reference_parity(sd15_dir, tmp_path):
    """The plan bench.py TIMES: hip_autotune = 1 seeded from the shipped table onnxstream_amd/tune/mi355x.txt (other tiles, ring depths, split-K, two wave
    groups on 12 of 62 shapes than the deterministic cost-model plan the other tests run).  A process of its own (the table is process-wide and read once);
    eager == captured == replayed bit for bit, then the triangulated bound against the reference's own fp16 / fp32 passes (missing #3)."""
    import shutil
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = str(tmp_path / "tune.txt")
    shutil.copy(os.path.join(repo, "onnxstream_amd", "tune", "mi355x.txt"), table)
    out = str(tmp_path / "tuned.npy")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_fullsize as t; from onnxstream_amd import build as b; "
            "from onnxstream_amd.synth import sd_unet; a, c = sd_unet.unet_inputs(sd_unet.SD15, 42), sd_unet.unet_inputs(sd_unet.SD15, 43); "
            "o = t._run(b.LIB_HOST, sys.argv[1], [a, c], runs=3, options=(('hip_autotune', 1),)); "
            "assert all(np.array_equal(o[0][0], x[0]) and np.array_equal(o[0][1], x[1]) for x in o[1:]), 'tuned plan: eager / captured / replayed differ'; "
            "np.save(sys.argv[2], o[0][0])" % (repo, os.path.dirname(os.path.abspath(__file__))))
    subprocess.check_call([sys.executable, "-c", code, sd15_dir, out], env=dict(os.environ, OSG_TUNE_CACHE=table))
    got = np.load(out)
    assert np.isfinite(got).all()
    if not oref.available():
        pytest.skip("oracle/_ref not present: reproducibility checked, reference parity skipped")
    _triangulated(got, sd15_dir, sd_unet.unet_inputs(sd_unet.SD15, 42), "SD1.5 UNet full size, tuned plan (shipped table)", "sd15")


def test_shipped_tune_table_covers_the_headline_plan(sd15_dir, tmp_path):
    """N = 1 and N > 1 must time the SAME plan (round 5): the ranks of an N > 1 job seed the shipped table onnxstream_amd/tune/mi355x.txt and run OSG_TUNE_FROZEN = 1 --
    a shape the table does not hold would take the cost model's first candidate there and be measured live at N = 1.  So: the full-size SD 1.5 UNet (cond + uncond,
    as bench.py pushes them) and the SD VAE decoder planned under the frozen shipped table leave osg_tune_misses() at 0."""
    import shutil
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = str(tmp_path / "tune.txt")
    shutil.copy(os.path.join(repo, "onnxstream_amd", "tune", "mi355x.txt"), table)
    vae = _vae_dir("sd_vae")
    code = ("import sys, ctypes, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_fullsize as t; from onnxstream_amd import build as b; "
            "from onnxstream_amd.bindings import Model; from onnxstream_amd.synth import sd_unet, sd_vae; "
            "a, c = sd_unet.unet_inputs(sd_unet.SD15, 42), sd_unet.unet_inputs(sd_unet.SD15, 43); "
            "t._run(b.LIB_HOST, sys.argv[1], [a, c], runs=1, options=(('hip_autotune', 1),)); "
            "m = Model(b.LIB_HOST, 0, 'ram+nocache'); m.read_file(sys.argv[2] + 'model.txt'); m._set_option('hip_autotune', 1); "
            "[m.add_tensor(k, v) for k, v in sd_vae.vae_inputs(sd_vae.SD_VAE).items()]; m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True); m.run(); m.close(); "
            "n = ctypes.CDLL(b.LIB_GPU).osg_tune_misses(); print('tune table misses:', n); sys.exit(3 if n else 0)" % (repo, os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code, sd15_dir, vae], env=dict(os.environ, OSG_TUNE_CACHE=table, OSG_TUNE_FROZEN="1", OSG_TUNE_LOG_MISSES="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print(r.stdout[-3000:])
    assert r.returncode == 0, "the shipped tune table does not hold every shape of the SD 1.5 UNet + VAE plan (or the run failed): see the [tune] miss lines above"


@pytest.fixture(scope="module")
def sd15_w8_dir():
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), "sd15_w8") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_unet.build_unet(DirSink(d), sd_unet.SD15, quant_weights=True)
        open(d + ".complete", "w").write("ok")
    return d


@pytest.mark.parametrize("resident", [0, 1])
def test_sd15_unet_w8a16_reference_parity_full_size(sd15_w8_dir, resident):
    """BASELINE config 3's UNet half at full size: uint8 weights with per-tensor (scale, zero point) in model.txt, fp16 activations.  The reference
    dequantises at load (get_tensor_data, src/onnxstream.cpp:2887-2891, dequantize :3353), w = f16((float)(q - zp) * scale) -- so does the default plan;
    hip_w8_resident keeps the CODES in HBM and turns them into halves between the LDS tile and the MFMA of the tuned kernels (osg_gemm_w8.hip, round 6): the exact
    integer q - zp into the MFMA, the scale on the f32 accumulator -- no per-weight rounding, the closer of the two to the fp32 output.  Both against the reference on
    the SAME quantised model directory (missing #2)."""
    from onnxstream_amd import build as b
    a, c = sd_unet.unet_inputs(sd_unet.SD15, 42), sd_unet.unet_inputs(sd_unet.SD15, 43)
    o = _run(b.LIB_HOST, sd15_w8_dir, [a, c], runs=2, options=(("hip_w8_resident", resident),))
    assert np.array_equal(o[0][0], o[1][0]) and np.array_equal(o[0][1], o[1][1])
    assert np.isfinite(o[0][0]).all()
    if not oref.available():
        pytest.skip("oracle/_ref not present: reproducibility checked, reference parity skipped")
    _triangulated(o[0][0], sd15_w8_dir, a, f"SD1.5 UNet full size, W8A16 (hip_w8_resident={resident})", "sd15_w8")


@pytest.fixture(scope="module")
def sdxl_dir():
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), "sdxl") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_unet.build_unet(DirSink(d), sd_unet.SDXL)
        open(d + ".complete", "w").write("ok")
    return d


def test_sdxl_unet_reference_parity_full_size(sdxl_dir):
    """BASELINE config 4 at full size: the SDXL-base UNet (2.57 B parameters, 2x4x128x128 latents, context 77x2048, 70 transformer blocks,
    linear proj_in/out, the added time_ids / text_embeds embedding) -- cond + uncond as one batch-2 pass against the reference's own
    fp16 and fp32 CPU passes on the same inputs (oracle/_ref on the GPU box's host), and bitwise reproducibility eager / captured."""
    from onnxstream_amd import build as b
    a, c = sd_unet.unet_inputs(sd_unet.SDXL, 42), sd_unet.unet_inputs(sd_unet.SDXL, 43)
    both = _run(b.LIB_HOST, sdxl_dir, [a, c], runs=2)
    assert np.array_equal(both[0][0], both[1][0]) and np.array_equal(both[0][1], both[1][1])
    assert np.isfinite(both[0][0]).all() and np.isfinite(both[0][1]).all()
    if not oref.available():
        pytest.skip("oracle/_ref not present: reproducibility checked, reference parity skipped")
    _triangulated(both[0][0], sdxl_dir, a, "SDXL UNet full size", "sdxl")


# ---- the VAE decoder at BASELINE's full size ([1,4,64,64] -> [1,3,512,512]): it runs inside the headline's timed region (fp16) and is
# ---- BASELINE config 3's W8A8 half (uint8); reference src/sd.cpp:1174-1256 (decoder_solver), :1212-1222 (m_use_uint8_arithmetic) -------------
def _vae_dir(name, **kw):
    from onnxstream_amd.synth import sd_vae
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), name) + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_vae.build_vae_decoder(DirSink(d), sd_vae.SD_VAE, **kw)
        open(d + ".complete", "w").write("ok")
    return d


def test_sd_vae_decoder_fp16_reference_parity_full_size():
    """fp16 arithmetic: 128 -> 128 channel convolutions on 512 x 512 pixels (the largest activations of the whole pipeline, 67 MB), the 512-wide
    single-head attention over 4 096 tokens (head dim > 160: the reference's own unfused MatMul / Mul / Softmax / MatMul sequence), 3 nearest
    upsamples.  Bound: the triangulated whole-net bound of tests/test_golden.py; plus eager == captured bit for bit."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_vae
    d = _vae_dir("sd_vae")
    z = sd_vae.vae_inputs(sd_vae.SD_VAE)
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m.read_file(d + "model.txt")
    outs = []
    for _ in range(2):
        for k, v in z.items():
            m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        m.run()
        outs.append(m.get_tensor("out_image")[0])
        m.clear_tensors()
    m.close()
    assert outs[0].shape == (1, 3, 512, 512) and np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])
    if not oref.available():
        pytest.skip("oracle/_ref not present: reproducibility checked, reference parity skipped")
    _triangulated(outs[0], d, z, "SD VAE decoder full size (fp16)", "vae", out="out_image", sub=(Ellipsis, slice(None, None, 4), slice(None, None, 4)))


def test_sd_vae_decoder_qu8_bit_exact_full_size():
    """uint8 arithmetic, the reference's W8A8 mode on the FULL-SIZE decoder with the shipped range data (onnxstream_amd/synth/data: a calibration
    pass of this very seeded model): the fp32 image the device returns must equal the reference's bit for bit, i.e. every final code -- and
    hence every code of the 60-odd requantising ops upstream on up to 512 x 512 x 128 tensors -- is the reference's."""
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_vae
    if not oref.available():
        pytest.skip("oracle/_ref not present")
    d = _vae_dir("sd_vae_qu8", quant_all=True)
    ranges = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "onnxstream_amd", "synth", "data", "sd_vae_qu8_range_data.txt"), newline="").read()
    open(d + "range_data.txt", "w", newline="").write(ranges)
    z = sd_vae.vae_inputs(sd_vae.SD_VAE)
    threads = oref.usable_cores()                      # the chunking of the pushed input's percentiles follows the thread count (reference :3091-3104): same on both sides
    m = Model(b.LIB_HOST, threads, "ram+nocache")
    m.hip_read_range_data(d + "range_data.txt")
    m.set_use_uint8_arithmetic(True)
    m.read_file(d + "model.txt")
    outs = []
    for _ in range(2):                                 # eager, then the captured graph behind the dynamically quantised input
        m.add_tensor(sd_vae.SD_VAE.in_name, z[sd_vae.SD_VAE.in_name])
        m.run()
        outs.append(m.get_tensor("out_image")[0])
        m.clear_tensors()
    m.close()
    want = oref.run_model_u8(d, z, ranges, threads=threads)["out_image"]
    assert outs[0].shape == want.shape == (1, 3, 512, 512)
    assert len(np.unique(want)) > 64                   # a real image: the code range is exercised
    assert np.array_equal(outs[0], want), int((outs[0] != want).sum())
    assert np.array_equal(outs[1], want), int((outs[1] != want).sum())
