"""Dev probe (GPU box, OSG_KDBG=1), round 6: the k loop of gemm2_kernel on the hot GEMM shapes of a batch-2 SD 1.5 pass, one line per
(shape, tile, ring, form): where ONE workgroup's time goes (entry, prologue requests, first tile, k loop, epilogue, store drain -- the stamps of
tools/kernel_phase_probe.py), and for the k loop itself
    cycles per 64-deep k-tile (at the nominal 2.4 GHz),  delivered B/clk per CU = (BM + BN) x 128 B / those cycles,  matrix-pipe use = BM x BN x 128 / 4069 / those cycles
so that a change to the loop can be judged without a pass.  Operands cold (384 MiB fill before every launch, as inside a pass) unless PROBE_HOT=1.
usage: gemm_kloop_probe.py [quick]"""
import ctypes, os, sys
os.environ["OSG_KDBG"] = "1"
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import osgpu

g = osgpu.Gpu(0)
L = g.lib
L.osg_kdbg_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
f16 = np.float16
DT16 = osgpu._NP2DT[np.dtype(f16)]
rng = np.random.default_rng(0)
evict = g.empty((384 << 20,), np.uint8)
TILES = {0: (128, 128), 1: (128, 64), 2: (64, 64), 3: (64, 128), 4: (128, 160), 5: (128, 80), 6: (64, 80), 7: (64, 160)}
GHZ = 2.4


def phases(launch, wgs, reps=5):
    rows = []
    for r in range(reps + 1):
        if not os.environ.get("PROBE_HOT"):
            g._ck(L.osg_memset(g.ctx, evict.ptr, r & 255, evict.nbytes))
        g.sync()
        g.timer_start()
        launch()
        ms = g.timer_stop()
        buf = np.zeros((wgs, 8), np.int64)
        g._ck(L.osg_kdbg_read(g.ctx, buf.ctypes.data, buf.nbytes))
        if r:
            rows.append((buf, ms))
    a = np.stack([b for b, _ in rows]).astype(np.float64)   # [reps, wgs, 8], 10 ns ticks
    t0 = a[:, :, 0].min(axis=1, keepdims=True)
    rel = (a - t0[:, :, None]) * 0.01
    med = np.median(rel, axis=(0, 1))
    last = np.median(rel[:, :, 6].max(axis=1))
    return med, np.diff(med[:7]), last, float(np.median([m for _, m in rows])) * 1e3


def gemm_case(M, N, K, cfg, nst, splits=1, ks=1, spec=0, fold=0, note="", batch=1):
    A = g.to_dev((rng.standard_normal((M, K)) * 0.5).astype(f16))
    W = g.to_dev((rng.standard_normal((N, K)) * K ** -0.5).astype(f16))
    Y = g.empty((batch * M, N), f16)
    bias = g.to_dev(np.zeros(N, f16))
    for k, v in (("OSG_GEMM_CFG", cfg), ("OSG_GEMM_NST", nst), ("OSG_GEMM_SPLITS", splits), ("OSG_GEMM_KS", ks), ("OSG_GEMM_SPEC", spec), ("OSG_GEMM_FOLD", fold)):
        os.environ[k] = str(v)
    bm, bn = TILES[cfg]
    wgs = -(-M // bm) * -(-N // bn) * splits * batch
    bptr = None if os.environ.get("PROBE_NO_BIAS") else bias.ptr     # (PROBE_NO_BIAS=1: what the on-demand bias load of the 128-row tiles costs in the epilogue phase)
    med, seg, last, us = phases(lambda: g._ck(L.osg_gemm(g.ctx, DT16, A.ptr, W.ptr, 1, bptr, DT16, None, Y.ptr, M, N, K, batch, 0, 0, M * N, 0)), wgs)
    nkt = max(1, (K // 64 + splits - 1) // splits) // (2 if ks == 2 else 1)
    cyc = seg[2] * 1e3 * GHZ / max(nkt, 1)
    bnp = (bn + 31) // 32 * 32
    bytes_kt = (bm + bn) * 128 * (2 if ks == 2 else 1)
    mfma_cyc = bm * bn * 128 / 4069.0 * (2 if ks == 2 else 1)
    print(f"GEMM {M:5d}x{N:5d}x{K:5d} tile {bm:3d}x{bn:<3d} ring {nst} splits {splits}{' KS2' if ks == 2 else ''}{' +loader waves' if spec else ''}{' fold' if fold else ''} {note:14s} wgs {wgs:5d} ({wgs / 256:4.2f}/CU) | "
          f"entry +{med[0]:4.2f} issue {seg[0]:4.2f} first {seg[1]:4.2f} k-loop {seg[2]:6.2f} ({nkt:2d} k-tiles: {cyc:5.0f} clk each, {bytes_kt / cyc:5.1f} B/clk/CU, matrix pipe {100 * mfma_cyc / cyc:4.1f} %) "
          f"epi {seg[3] + seg[4]:4.2f} drain {seg[5]:4.2f} | median wg {med[6]:6.2f} last wg {last:6.2f} launch {us:6.2f} us = {2.0 * M * N * K / us / 1e6:5.0f} TFLOP/s", flush=True)
    for b in (A, W, Y, bias):
        b.free()


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
pitch = len(sys.argv) > 1 and sys.argv[1] == "pitch"   # round 6: is the k loop's delivery rate a matter of the operands' row pitch (L2 channel / DRAM bank camping)?
print(f"# {'hot' if os.environ.get('PROBE_HOT') else 'cold'} operands; cycles at {GHZ} GHz nominal; B/clk/CU and matrix-pipe use are of the k loop of the MEDIAN workgroup")
# (shape, [configs]): the round-5 table's choice first, then the round-6 tiles / rings
cases = [
    ((512, 10240, 1280), [(0, 2, 1, 1, 0), (0, 4, 1, 1, 0), (4, 2, 1, 1, 0), (4, 4, 1, 1, 0), (4, 4, 1, 1, 1), (0, 4, 1, 1, 1)]),
    ((2048, 5120, 640), [(1, 2, 1, 1, 0), (0, 2, 1, 1, 0), (4, 2, 1, 1, 0), (4, 4, 1, 1, 0), (4, 4, 1, 1, 1)]),
    ((2048, 640, 2560), [(2, 4, 1, 1, 0), (6, 4, 1, 1, 0), (6, 6, 1, 1, 0), (5, 4, 1, 1, 0)]),
    ((512, 1280, 5120), [(3, 4, 2, 1, 0), (5, 4, 4, 1, 0), (6, 4, 2, 1, 0)]),
    ((8192, 960, 320), [(0, 2, 1, 1, 0), (4, 2, 1, 1, 0), (4, 4, 1, 1, 0)]),
    ((2048, 1920, 640), [(0, 4, 1, 1, 0), (0, 4, 1, 1, 1), (7, 2, 1, 1, 0), (4, 2, 1, 1, 0)]),
    ((2048, 640, 640), [(2, 4, 1, 1, 0), (2, 2, 1, 2, 0), (6, 4, 1, 1, 0)]),
    ((512, 1280, 1280), [(2, 4, 1, 1, 0), (2, 4, 1, 2, 0), (6, 4, 1, 1, 0)]),
    ((8192, 320, 320), [(2, 2, 1, 1, 0), (5, 4, 1, 1, 0), (5, 2, 1, 1, 0)]),
]
if len(sys.argv) > 1 and sys.argv[1] == "ideal":
    # round 6: every workgroup the SAME tile (batch items broadcast from one A and one B): the k loop with an ideal memory system behind it
    for (M, N, K, cfg, nst) in ((64, 64, 640, 2, 4), (64, 64, 2560, 2, 4), (64, 80, 640, 6, 4), (128, 128, 1280, 0, 4), (128, 128, 1280, 0, 2), (128, 160, 1280, 4, 4), (128, 64, 640, 1, 4), (128, 80, 640, 5, 4)):
        gemm_case(M, N, K, cfg, nst, note="256 x the same tile", batch=256)
    g.close()
    sys.exit(0)
if pitch:
    cases = [((M, N, K + dk), cfgs[:2]) for (M, N, K), cfgs in (((512, 10240, 1280), [(4, 4, 1, 1, 0), (0, 4, 1, 1, 0)]), ((2048, 640, 640), [(2, 4, 1, 1, 0), (6, 4, 1, 1, 0)]),
                                                                ((2048, 640, 2560), [(2, 4, 1, 1, 0), (6, 4, 1, 1, 0)]), ((512, 1280, 1280), [(2, 4, 1, 1, 0), (6, 4, 1, 1, 0)]),
                                                                ((2048, 1920, 640), [(0, 4, 1, 1, 0), (4, 2, 1, 1, 0)]), ((8192, 320, 320), [(5, 4, 1, 1, 0), (2, 2, 1, 1, 0)]))
             for dk in (0, 64, 192)]
for (M, N, K), cfgs in (cases[:3] if quick else cases):
    for cfg, nst, splits, ks, spec in cfgs:
        gemm_case(M, N, K, cfg, nst, splits, ks, spec)
g.close()
