"""ScaledDotProductAttention cases (first half): llama-style attention blocks whose score chains are the two
forms the reference rewrites into its ScaledDotProductAttention pseudo-op when m_use_scaled_dp_attn_op is set
(src/onnxstream.cpp:3635-3755): Transpose/MatMul/Div/Add/Softmax/MatMul and Transpose/Mul/Mul/MatMul/Add/Softmax/MatMul.

The oracle build (torch's XNNPACK) has no xnn_*_scaled_dot_product_attention_* entry points, so the reference cannot execute the
fused op here; it runs the SAME chains unfused through genuine XNNPACK operators, and that output is the fixture
(tools/make_golden_sdpa.py -> tests/golden/sdpa_*.npz).  The HIP backend is tested against it both ways: flag off (op by op) and
flag on (fused kernel)."""
from __future__ import annotations

import numpy as np

from onnxstream_amd.synth.graph import GraphBuilder

f32 = np.float32


def _rn(seed, shape, std=1.0):
    return (np.random.default_rng(seed).standard_normal(shape, dtype=f32) * f32(std)).astype(f32)


def _causal(tq, tk, neg):
    m = np.zeros((1, 1, tq, tk), f32)
    past = tk - tq
    for i in range(tq):
        m[0, 0, i, past + i + 1:] = neg
    return m


def _block(g: GraphBuilder, x, ctx, heads, form, neg=-65504.0):
    """x:[1,T,C] queries, ctx:[1,S,C] keys/values (S >= T: S - T cached tokens precede the T new ones)."""
    _, tq, c = x.shape
    _, tk, _ = ctx.shape
    d = c // heads

    def split(nm, t, tokens):
        r = g.reshape(f"/attn/{nm}/Reshape", t, (1, tokens, heads, d))
        return g.transpose(f"/attn/{nm}/Transpose", r, (0, 2, 1, 3))          # [1,H,tokens,d]

    q = split("q", g.matmul_w("/attn/q_proj", x, c), tq)
    k = split("k", g.matmul_w("/attn/k_proj", ctx, c), tk)
    v = split("v", g.matmul_w("/attn/v_proj", ctx, c), tk)
    mask = g.weight("attn.mask", _causal(tq, tk, neg), allow_quant=False, q8_exempt=True)
    kt = g.transpose("/attn/Transpose_3", k, (0, 1, 3, 2))
    if form == "div":
        s = g.op("/attn/MatMul", "MatMul", [q, kt], (1, heads, tq, tk))
        s = g.binary("/attn/Div", "Div", s, g.scalar("attn.sqrt_d", float(np.sqrt(d))))
    else:
        q2 = g.binary("/attn/Mul", "Mul", q, g.scalar("attn.s", float(d ** -0.25)))
        k2 = g.binary("/attn/Mul_1", "Mul", kt, g.scalar("attn.s2", float(d ** -0.25)))
        s = g.op("/attn/MatMul", "MatMul", [q2, k2], (1, heads, tq, tk))
    s = g.binary("/attn/Add", "Add", s, mask)
    p = g.op("/attn/Softmax", "Softmax", [s], s.shape, {"axis": "-1"})
    o = g.op("/attn/MatMul_1", "MatMul", [p, v], (1, heads, tq, d))
    o = g.transpose("/attn/Transpose_4", o, (0, 2, 1, 3))
    o = g.reshape("/attn/Reshape_3", o, (1, tq, c))
    return g.linear("/attn/o_proj", o, c, bias=False)


def sdpa_div(g):          # prefill: T = S = 24, 4 heads of 16, causal mask
    x = g.input("x", (1, 24, 64))
    _block(g, x, x, 4, "div")
    return {"x": _rn(31, (1, 24, 64))}


def sdpa_mulmul(g):       # the Mul/Mul form, decode-like: 5 new tokens over 37 keys (32 cached), 2 heads of 64
    x = g.input("x", (1, 5, 128))
    c = g.input("ctx", (1, 37, 128))
    _block(g, x, c, 2, "mulmul")
    return {"x": _rn(32, (1, 5, 128)), "ctx": _rn(33, (1, 37, 128))}


def sdpa_long(g):         # more than one 64-key tile and a ragged tail: T = S = 150, 2 heads of 32
    x = g.input("x", (1, 150, 64))
    _block(g, x, x, 2, "div")
    return {"x": _rn(34, (1, 150, 64))}


CASES = [sdpa_div, sdpa_mulmul, sdpa_long]


def emit(case, sink, seed=4321):
    g = GraphBuilder(sink, seed=seed)
    ins = case(g)
    head, outp = g.lines[-1].split("*output:", 1)
    tok, _, rest = outp.partition("*")
    g.lines[-1] = head + "*output:out" + tok[tok.index("("):] + (("*" + rest) if rest else "")
    g.finish()
    return ins
