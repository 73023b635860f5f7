"""The parity bound of the whole-network / full-size tests: ONE definition, used by tests/test_golden.py (whole miniature nets,
real-width chains), tests/test_fullsize.py, __graft_entry__.smoke() and tools/golden_table.py.

    err16 = max|got - ref16| / max|ref32|      distance to the reference's fp16 output (the parity target of north_star: <= 1e-3)
    err32 = max|got - ref32| / max|ref32|      distance to the reference's fp32 output
    drift = max|ref16 - ref32| / max|ref32|    how far the reference's OWN fp16 path is from its fp32 path on this input

The reference's fp16 output is not one tensor per input: XNNPACK picks its micro-kernels per CPU, and the same model on the same input moves between hosts
(ref32 does not: bit-identical on every host seen).  Where the fixtures hold the fp16 output of SEVERAL hosts, err16 is the distance to the NEAREST of them, drift
the largest of theirs, and

    spread = max over host pairs of max|ref16_A - ref16_B| / max|ref32|      how far the reference is from ITSELF on this input -- a measured, committed number

A result passes when one of these holds:
  (a) err16 <= 1e-3                  it sits on the reference's fp16 output;
  (b) err32 <= drift                 it is as close to the fp32 truth as the reference's fp16 path gets ("closer to fp32 than the fp16 oracle is not
                                     an error");
  (p) err16 <= max(spread, 1e-3)     PINNED host-to-host leg (round 6; it replaces round 5's `err16 <= 0.2 x drift`, a ratio of a large number that left the
                                     full-size UNets 1.6-2.8x of head room): the device is no further from the nearest host's fp16 output than two hosts of the
                                     reference are from each other.  Only where at least two hosts' outputs are committed: the full-size nets
                                     (tests/golden/ref16_fullsize_{xeon,epyc}.npz, tools/ref16_fullsize.py: SD 1.5 spread 2.52e-3, W8A16 2.42e-3, SDXL: see the
                                     file) and the miniature nets of tests/golden/ref16_host2.npz.
Named exceptions carry their MEASURED number: in leg (b) a factor on drift, in leg (p) a factor on the spread -- measured ratio + <= 10 %, never a rounded-up bound.
"""
from __future__ import annotations

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# case key -> (factor on drift in leg (b), the measured err32 / drift that made it necessary, where it was measured)
EXCEPTIONS: dict = {
    # deterministic plans on MI355X, profiles/r05_golden_table.txt (fixture drift / second host's drift; the larger one is the bound's)
    "vae_tiny@f0": (1.10, 1.037, "err32 1.81e-3 vs drift 1.745e-3: one rounding point per graph op, 3 resolutions of GroupNorm + attention"),
    "unet_tiny_w8@f0": (1.55, 1.457, "err32 6.67e-3 vs drift 4.58e-3: the W8A16 miniature UNet at one kernel per graph op"),
    "unet_tiny_w8@f2": (1.05, 1.0002, "err32 4.580e-3 vs drift 4.579e-3"),
    "yolov8n scores": (1.25, 1.183, "err32 9.14e-3 vs drift 7.73e-3 (class scores in [0, 1] behind a sigmoid; fusion 0 and 2 alike; profiles/r05_parity_table.txt)"),
    "unet_tiny@tuned": (1.10, 1.017, "a timing-dependent plan: err32 seen between 4.6e-3 and 6.3e-3 across runs against a drift of 6.195e-3 (round 2)"),
}
# case key -> (factor on the host-to-host spread in leg (p), the measured err16 / spread, where it was measured); filled from profiles/r06_parity_table.txt
SPREAD_EXCEPTIONS: dict = {
    "SDXL UNet full size": (1.30, 1.181, "err16 3.33e-3 to the nearer host (EPYC) vs a spread of 2.82e-3 between Xeon and EPYC; err32 / drift 1.02 (profiles/r06_parity_table.txt)"),
}


def margins(err16: float, err32: float, drift: float, key: str | None = None, spread: float | None = None):
    f = EXCEPTIONS.get(key, (1.0,))[0] if key else 1.0
    m = {"a": err16 / 1e-3, "b": err32 / (f * drift) if drift > 0 else float("inf")}
    if spread is not None:
        fp = SPREAD_EXCEPTIONS.get(key, (1.0,))[0] if key else 1.0
        m["p"] = err16 / max(fp * spread, 1e-3)
    return m


def ok(err16: float, err32: float, drift: float, key: str | None = None, spread: float | None = None) -> bool:
    return min(margins(err16, err32, drift, key, spread).values()) <= 1.0


def describe(what: str, err16: float, err32: float, drift: float, key: str | None = None, spread: float | None = None) -> str:
    m = margins(err16, err32, drift, key, spread)
    leg = min(m, key=m.get)
    sp = f" host-to-host spread={spread:.2e}" if spread is not None else ""
    return (f"{what}: |gpu-ref16|/max={err16:.2e} |gpu-ref32|/max={err32:.2e} reference fp16 drift={drift:.2e}{sp} -> leg ({leg}) at {m[leg]:.2f} of its bound ["
            + ", ".join(f"{k} {v:.2f}" for k, v in m.items()) + "]")


def check(what: str, err16: float, err32: float, drift: float, key: str | None = None, spread: float | None = None):
    print(describe(what, err16, err32, drift, key, spread))
    assert ok(err16, err32, drift, key, spread), (what, err16, err32, drift, spread)


# ---- the reference's fp16 output on several hosts ---------------------------------------------------------------------------------------------------------
def fullsize_host_refs(case: str, ref32: np.ndarray, sub=None):
    """The committed fp16 outputs of the reference for one full-size case (`sd15`, `sd15_w8`, `sdxl`, `vae`), one per host file -- only those whose committed
    fp32 output equals `ref32` bit for bit (same model, same input: otherwise the fixture is about something else and is not used).  sub: the subsampling the
    fixture was stored with (the VAE image, [..., ::4, ::4]), applied to ref32 before the comparison."""
    out = []
    for host in ("xeon", "epyc"):
        p = os.path.join(GOLDEN, f"ref16_fullsize_{host}.npz")
        if not os.path.exists(p):
            continue
        z = np.load(p)
        if case + "_ref16" not in z.files:
            continue
        r32 = ref32[sub] if sub is not None else ref32
        if z[case + "_ref32"].shape == r32.shape and np.array_equal(z[case + "_ref32"], r32):
            out.append(z[case + "_ref16"].astype(np.float32))
    return out


def triangulate(got: np.ndarray, refs16: list, ref32: np.ndarray):
    """err16 (nearest host), err32, drift (largest host), spread (largest host pair; None with fewer than two hosts) of `got` against the reference's outputs."""
    mx = float(np.abs(ref32).max())
    err16 = min(float(np.abs(got - r).max()) for r in refs16) / mx
    err32 = float(np.abs(got - ref32).max()) / mx
    drift = max(float(np.abs(r - ref32).max()) for r in refs16) / mx
    spread = None
    if len(refs16) >= 2:
        spread = max(float(np.abs(refs16[i] - refs16[j]).max()) for i in range(len(refs16)) for j in range(i + 1, len(refs16))) / mx
    return err16, err32, drift, spread
