"""Dev probe (GPU box): the fused transformer-block tail (osg_tblock_tail) against the same chain as separate launches, at the SD 1.5 64x64 level
(M = 8192, C = 320), cycling through enough weight sets that every launch finds its weights cold (as inside a pass)."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from onnxstream_amd import osgpu
import test_tblock_tail as t

f16 = np.float16


def main():
    gpu = osgpu.Gpu(0)
    C, heads, M, imgs, Tk = 320, 8, 8192, 2, 77
    nsets = int(os.environ.get("NSETS", "96"))
    reps = int(os.environ.get("REPS", "3"))
    rows = int(os.environ.get("ROWS", "0"))        # rows per block: 0 = the library's choice, 32, 64
    print(f"rows per block: {rows or 'library choice'}")
    rng = np.random.default_rng(0)
    sets = []
    w0 = t.make_block(rng, C)
    for i in range(nsets):
        sets.append(gpu.tblock_weights({n: np.roll(v, i, axis=0) for n, v in w0.items()}))
    a1, x0, xin = gpu.to_dev(t.rnd(rng, (M, C))), gpu.to_dev(t.rnd(rng, (M, C))), gpu.to_dev(t.rnd(rng, (M, C)))
    k, v = gpu.to_dev(t.rnd(rng, (imgs, Tk, C))), gpu.to_dev(t.rnd(rng, (imgs, Tk, C)))
    kp, vtp = gpu.tblock_kv_pack(k, v, heads)
    scale = 40 ** -0.5
    out = gpu.empty((M, C), f16)
    for proj in (True, False):
        for name, ns in (("cold", nsets), ("hot", 1)):
            for _ in range(2):
                gpu.tblock_tail(a1, x0, sets[0] if proj else {**sets[0], 'wpo': None, 'bpo': None}, kp, vtp, Tk, heads, scale, M // imgs, xin=xin if proj else None, out=out, rows_per_block=rows)
            gpu.sync()
            gpu.timer_start()
            n = 0
            for r in range(reps):
                for i in range(ns if ns > 1 else 32):
                    w = dict(sets[i % ns])
                    if not proj:
                        w["wpo"] = w["bpo"] = None
                    gpu.tblock_tail(a1, x0, w, kp, vtp, Tk, heads, scale, M // imgs, xin=xin if proj else None, out=out, rows_per_block=rows)
                    n += 1
            ms = gpu.timer_stop()
            print(f"fused tail proj_out={int(proj)} {name}: {1000 * ms / n:.1f} us per launch ({n} launches)")
    # ---- where a row block's time goes: wall-clock stamps of every stage (dbg[7]), cold weights
    st = gpu.to_dev(np.zeros((M // 32, 32), np.int64))
    names = ["rows landed", "to_out1", "LN", "to_q", "cross-attn", "to_out2", "LN", "GEGLU chunk 0", "chunks 1..9 (+ff2 0..8)", "ff2 chunk 9", "x3 -> LDS", "proj_out", "stores drained"]
    acc = np.zeros(13)
    inner = np.zeros(15)
    nrep = 8
    for i in range(nrep):
        gpu.tblock_tail(a1, x0, sets[(7 * i + 3) % nsets], kp, vtp, Tk, heads, scale, M // imgs, xin=xin, out=out, stamps=st, rows_per_block=rows)
        s = st.numpy().astype(np.float64)
        s = s[s[:, 0] != 0]                                    # (M / 32 slots: 64-row blocks fill the first half)
        acc += (s[:, 1:14] - s[:, 0:13]).mean(0) / 100.0      # 100 MHz -> us
        tot = (s[:, 13] - s[:, 0]) / 100.0
        inner += (s[:, 17:32] - s[:, 16:31]).mean(0) / 100.0
        span = (s[:, 13].max() - s[:, 0].min()) / 100.0
    print("stage (us, mean over the row blocks and %d launches):" % nrep)
    for n, v in zip(names, acc / nrep):
        print(f"  {n:28s} {v:7.2f}")
    inames = ["to_out1 k-tile 0", "k-tile 1", "k-tile 2", "k-tile 3", "k-tile 4", "epilogue -> LDS", "(to chunk 2's barrier)", "chunk 3: ff.net.0.proj (5 k-tiles)", "chunk 3: ff.net.2 (2 k-tiles)", "chunk 3: GEGLU arithmetic + store", "chunk 3: barrier",
              "chunk 4: ff.net.0.proj", "chunk 4: ff.net.2", "chunk 4: GEGLU", "chunk 4: barrier"]
    for n, v in zip(inames, inner / nrep):
        print(f"    {n:36s} {v:7.2f}")
    print(f"  row block entry -> drained: mean {tot.mean():.1f}, min {tot.min():.1f}, max {tot.max():.1f}; first entry -> last drained {span:.1f} us (last launch)")
    if os.environ.get("SKIP_SEP", "1" if rows else ""):
        return
    # the separate chain, same weights (hot and cold)
    q3shape = (imgs, M // imgs, C)
    for name, ns in (("cold", nsets), ("hot", 1)):
        gpu.sync()
        gpu.timer_start()
        n = 0
        for r in range(reps):
            for i in range(ns if ns > 1 else 32):
                d = sets[i % ns]
                x1 = gpu.gemm(a1, d["wo1"], d["bo1"], x0, b_is_nk=True)
                n2 = gpu.layer_norm(x1, d["g2"], d["be2"], 1e-5)
                q = gpu.gemm(n2, d["wq2"], None, None, b_is_nk=True)
                q.shape = q3shape
                a2 = gpu.attention_tokens(q, k, v, heads, scale)
                a2.shape = (M, C)
                x2 = gpu.gemm(a2, d["wo2"], d["bo2"], x1, b_is_nk=True)
                n3 = gpu.layer_norm(x2, d["g3"], d["be3"], 1e-5)
                h = gpu.gemm(n3, d["w1"], d["b1"], None, b_is_nk=True)
                hg = gpu.geglu(h)
                x3 = gpu.gemm(hg, d["w2"], d["b2"], x2, b_is_nk=True)
                y = gpu.gemm(x3, d["wpo"], d["bpo"], xin, b_is_nk=True)
                n += 1
        ms = gpu.timer_stop()
        print(f"separate launches (10, unfused LayerNorm / GEGLU) {name}: {1000 * ms / n:.1f} us per chain")


if __name__ == "__main__":
    main()
