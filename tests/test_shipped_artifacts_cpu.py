"""CPU-side guards on two shipped artefacts that the GPU run depends on but no GPU is needed to check:

  * onnxstream_amd/tune/mi355x.txt -- every row must survive the loader's validation (osg_ctx.hip `load_locked`: a row that names no launchable configuration
    is DROPPED silently, and the shape would then be re-timed on the box, or -- frozen, the N > 1 ranks -- run the cost model's first candidate while the
    bench line still says "shipped table").  The validation is restated here from that function; keys must be unique; the headline's rows come first.
  * onnxstream_amd/libosgpu.so -- register / scratch budget of the hot kernels, read from the code objects' metadata (tools/kernel_resources.py): a kernel of the
    contraction / attention / normalisation families that starts spilling to scratch is a performance regression the parity tests cannot see.
"""
import os
import shutil
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(REPO, "onnxstream_amd", "tune", "mi355x.txt")


def _rows():
    out = []
    for ln, line in enumerate(open(TABLE).read().split("\n"), 1):
        if not line.strip():
            continue
        f = line.split()
        assert len(f) == 19, f"line {ln}: {len(f)} fields (the loader reads 18 ints and a float)"
        out.append((ln, [int(x) for x in f[:18]], float(f[18])))
    return out


def _launchable(family, cfg, nst, splits, bn):
    """osg_ctx.hip load_locked(), restated: which (family, cfg, nst, splits, bn) name an instantiation the launchers have."""
    tile, ks2, fold, spec = cfg & 7, cfg & 8, cfg & 16, cfg & 32
    if family == 0:      # gemm2_kernel: cfg = tile | KS2 << 3 | fold << 4 | four loader waves << 5; tiles 4 .. 7 = the 160 / 80-column tiles
        if cfg < 0 or (cfg & ~63):
            return False
        if spec and not (tile in (0, 4) and nst == 4 and not ks2 and not fold and splits == 1):
            return False
        if fold and (ks2 or tile in (0, 4, 7) or not 2 <= splits <= 4):
            return False
        if ks2 and not ((tile == 2 and nst in (2, 4)) or (tile == 1 and nst == 2)):
            return False
        if tile <= 3:
            if not (nst in (2, 4) or (nst == 6 and tile >= 1) or (nst == 8 and tile == 2)):
                return False
        elif not (nst in (2, 4) or (nst == 6 and tile == 6)):
            return False
        return 1 <= splits <= 64
    if family == 1:      # conv3x3_kernel: cfg = fold << 4, nst = loader waves
        if not (cfg == 0 or (cfg == 16 and 2 <= splits <= 4)):
            return False
        return bn in (80, 128, 160) and 1 <= splits <= 64 and nst in (0, 4, 8)
    return False


def test_every_row_of_the_shipped_tune_table_survives_the_loader():
    rows = _rows()
    assert len(rows) >= 77
    seen = {}
    for ln, v, us in rows:
        kind, device, M, N, K, batch = v[:6]
        family, cfg, nst, splits, bn = v[13:18]
        assert kind in (0, 1, 2), f"line {ln}"
        assert device == 0, f"line {ln}: the device ordinal is not part of a shape's identity (one table serves every rank)"
        assert M > 0 and N > 0 and K > 0 and batch > 0, f"line {ln}"
        assert _launchable(family, cfg, nst, splits, bn), f"line {ln}: the loader would drop this row: family {family} cfg {cfg} nst {nst} splits {splits} bn {bn}"
        assert family == 0 or kind == 1, f"line {ln}: only a 3x3 / stride 1 convolution can take the halo-reuse kernel"
        assert us > 0, f"line {ln}: a shipped row carries the time that was measured for it"
        key = tuple(v[:13])
        assert key not in seen, f"lines {seen.get(key)} and {ln} name the same shape"
        seen[key] = ln


def test_the_headline_rows_of_the_tune_table_come_first():
    """Rows are appended as configurations are added (SDXL, 4 prompts per GPU: round 5); the first 77 are the SD 1.5 UNet + VAE decoder at batch 2, the plan
    bench.py times by default -- M of a batch-2 SD 1.5 pass never exceeds 2 x 64 x 64 rows for the UNet and 512 x 512 for the decoder."""
    rows = _rows()[:77]
    assert max(v[2] for _, v, _ in rows) <= 512 * 512
    assert sum(1 for _, v, _ in rows if v[13] == 1) >= 5          # the halo-reuse convolution wins several of the UNet's 3x3 shapes
    assert sum(1 for _, v, _ in rows if v[14] & 16) >= 1         # round 5: split-K folded in the kernel where it measured faster (3x3 convolutions at 32x32 / 64x64)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None, reason="needs llvm-readelf and c++filt")
def test_hot_kernels_do_not_spill_to_scratch():
    from onnxstream_amd import build as b
    if not os.path.exists(b.LIB_GPU):
        import __graft_entry__ as ge
        ge.build()
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import kernel_resources as kr
    blob = open(b.LIB_GPU, "rb").read()
    rows = []
    for _, obj in kr.code_objects(blob):
        if obj.startswith(b"\x7fELF"):
            rows += list(kr.kernels_of(obj))
    names = kr.demangle([r["name"] for r in rows])
    assert len(rows) > 200
    by = {}
    for r, n in zip(rows, names):
        by[n.replace("void ", "")] = r
    hot = [n for n in by if any(k in n for k in ("gemm2_kernel", "attn2_kernel", "attn_kernel", "tblock_tail_kernel", "gn_slab_kernel", "gn_apply",
                                                  "splitk_reduce", "layer_norm_kernel", "q8_gemm_kernel", "q8_conv"))]
    assert len(hot) > 100
    for n in hot:
        assert by[n]["scratch"] == 0 and by[n]["vspill"] == 0, f"{n}: scratch {by[n]['scratch']} B, {by[n]['vspill']} VGPR spills"
        assert by[n]["vgpr"] <= 512      # .vgpr_count is the unified total (AGPRs included)
    # the halo-reuse convolution: the variants with 4 loader waves (512 threads) are the ones every measured plan uses; they must be clean.  The 8-loader-wave
    # variants at BN >= 128 are known to spill (768 threads leave 168 registers per lane, profiles/r05_kernel_resources.txt): candidates the tuner never picked.
    conv = {n: r for n, r in by.items() if "conv3x3_kernel" in n}
    assert len(conv) >= 20
    for n, r in conv.items():
        loaders = int(n.split("<")[1].split(">")[0].split(",")[5])
        bn = int(n.split("<")[1].split(">")[0].split(",")[1])
        if loaders == 4 or bn == 80:
            assert r["scratch"] == 0 and r["vspill"] == 0, f"{n}: scratch {r['scratch']} B"
