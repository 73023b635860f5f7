"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's per-node arithmetic for the SD hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product (``onnxstream_amd``) never does and has no CPU fallback.

Each function restates what the reference computes for ONE graph op when ``m_use_fp16_arithmetic`` is on: f16 storage,
f32 (or double) math inside the op, one round-to-nearest-even to f16 on the op's output.  Citations are
``/root/reference/src/onnxstream.cpp`` line numbers.  The restatement is pinned against the real reference (built
unmodified into ``oracle/_ref``) by ``tests/test_golden.py (test_restatement_*)`` and the committed fixtures in ``tests/golden``.
"""
from __future__ import annotations

import math

import numpy as np

f16, f32, f64 = np.float16, np.float32, np.float64


def r16(x):
    """The single rounding every reference op applies on store (XNNPACK f16 kernels / push_tensor convert :3029)."""
    return np.asarray(x, dtype=f32).astype(f16)


# ---- dense contractions ------------------------------------------------------------------------------------------------
def conv2d_nhwc(x, w_ohwi, bias=None, stride=(1, 1), pads=(1, 1, 1, 1), residual=None):
    """XnnPack::convolution (:1292-1534): NHWC x OHWI -> NHWC, f32 accumulate, bias added in f32, one rounding.
    pads = (top, left, bottom, right) AFTER the reference's re-centring (:1315-1329)."""
    x = np.asarray(x)
    n, h, wd, cin = x.shape
    cout, kh, kw, _ = w_ohwi.shape
    pt, pl, pb, pr = pads
    sh, sw = stride
    xp = np.zeros((n, h + pt + pb, wd + pl + pr, cin), f32)
    xp[:, pt:pt + h, pl:pl + wd, :] = x.astype(f32)
    ho, wo = (h + pt + pb - kh) // sh + 1, (wd + pl + pr - kw) // sw + 1
    cols = np.empty((n, ho, wo, kh, kw, cin), f32)
    for i in range(kh):
        for j in range(kw):
            cols[:, :, :, i, j, :] = xp[:, i:i + sh * ho:sh, j:j + sw * wo:sw, :]
    acc = cols.reshape(n * ho * wo, kh * kw * cin).astype(f64) @ w_ohwi.reshape(cout, -1).astype(f64).T
    if bias is not None:
        acc = acc + np.asarray(bias).astype(f64)
    acc = acc.reshape(n, ho, wo, cout)
    if residual is not None:
        acc = acc + np.asarray(residual).astype(f64)
    return r16(acc)


def matmul(a, b_kn, bias=None, residual=None):
    """XnnPack::matrix_multiply (:1035-1215) / _dynamic (:929): [M,K]x[K,N], kernel [K,N] row-major, f32 accumulate."""
    acc = np.asarray(a).astype(f64) @ np.asarray(b_kn).astype(f64)
    if bias is not None:
        acc = acc + np.asarray(bias).astype(f64)
    if residual is not None:
        acc = acc + np.asarray(residual).astype(f64)
    return r16(acc)


# ---- elementwise ---------------------------------------------------------------------------------------------------------
def binary(kind, a, b):
    """XnnPack::add/subtract/multiply/divide (:1666,:1811,:846,:1881): numpy broadcasting, f32 math, one rounding."""
    a32, b32 = np.asarray(a).astype(f32), np.asarray(b).astype(f32)
    if kind == "add":
        return r16(a32 + b32)
    if kind == "sub":
        return r16(a32 - b32)
    if kind == "mul":
        return r16(a32 * b32)
    if kind == "div":
        return r16(a32 / b32)
    raise ValueError(kind)


def sigmoid(x):
    """XnnPack::sigmoid (:1217); XNNPACK's f16 vsigmoid is a polynomial approximation: agree to ~1 f16 ulp."""
    x32 = np.asarray(x).astype(f32)
    return r16(1.0 / (1.0 + np.exp(-x32.astype(f64))))


def silu(x):
    return binary("mul", x, sigmoid(x))


def erf(x):
    """Model::run Erf (:4001-4139): std::erf in f32 then one rounding."""
    v = np.asarray(x).astype(f64)
    return r16(np.vectorize(math.erf)(v))


def unary(kind, x, param=0.0):
    x32 = np.asarray(x).astype(f32)
    if kind == "sigmoid":
        return sigmoid(x)
    if kind == "erf":
        return erf(x)
    if kind == "sqrt":
        return r16(np.sqrt(x32))
    if kind == "sin":
        return r16(np.sin(x32.astype(f64)))
    if kind == "cos":
        return r16(np.cos(x32.astype(f64)))
    if kind == "neg":
        return r16(-x32)
    if kind == "pow":
        return r16(np.power(x32.astype(f64), param))   # std::pow in f32 (:5478-5604)
    if kind == "silu":
        return r16(x32.astype(f64) / (1.0 + np.exp(-x32.astype(f64))))
    if kind == "gelu_erf":
        v = x32.astype(f64)
        return r16(0.5 * v * (1.0 + np.vectorize(math.erf)(v / math.sqrt(2.0))))
    raise ValueError(kind)


# ---- normalisation / reductions ----------------------------------------------------------------------------------------
def instance_norm(x, scale, bias, eps):
    """Model::run InstanceNormalization (:4788-5055) on [1,G,L]: per row mean (double acc), variance of the deviations
    (double acc of float dev^2), y = scale*(x-mean)/sqrt(var+eps)+bias in f32 (:4935-4982), one rounding."""
    x64 = np.asarray(x).astype(f64)
    mean = x64.mean(axis=-1, keepdims=True)
    var = ((x64 - mean) ** 2).mean(axis=-1, keepdims=True)
    y = (x64 - mean) / np.sqrt(var + eps)
    g = x.shape[-2]
    if scale is not None:
        y = y * np.asarray(scale).astype(f64).reshape(g, 1)
    if bias is not None:
        y = y + np.asarray(bias).astype(f64).reshape(g, 1)
    return r16(y)


def reduce_mean_last(x):
    """Model::run ReduceMean (:5237-5393): last axis, keepdims, double accumulate, one rounding."""
    return r16(np.asarray(x).astype(f64).mean(axis=-1, keepdims=True))


def softmax_last(x):
    """XnnPack::softmax (:1958) -> XNNPACK three passes: e = exp(x - max) stored in f16, sum of the un-rounded e in
    f32, out = f16(e) * f16(1/sum)."""
    x32 = np.asarray(x).astype(f32)
    mx = x32.max(axis=-1, keepdims=True)
    e = np.exp((x32 - mx).astype(f64))
    s = e.sum(axis=-1, keepdims=True)
    rinv = r16(1.0 / s).astype(f32)
    return r16(r16(e).astype(f32) * rinv)


def layer_norm_decomposed(x, gamma, beta, eps):
    """The exported LayerNorm chain, one rounding per graph op (ReduceMean :5237, Sub :5394, Pow :5478, ReduceMean,
    Add, Sqrt :4001, Div :5605, Mul, Add)."""
    mean = reduce_mean_last(x)
    sub = binary("sub", x, mean)
    p = unary("pow", sub, 2.0)
    var = reduce_mean_last(p)
    ve = binary("add", var, r16(eps))
    sd = unary("sqrt", ve)
    d = binary("div", sub, sd)
    return binary("add", binary("mul", d, gamma), beta)


def layer_norm_exact(x, gamma, beta, eps):
    x64 = np.asarray(x).astype(f64)
    mean = x64.mean(-1, keepdims=True)
    var = ((x64 - mean) ** 2).mean(-1, keepdims=True)
    return r16((x64 - mean) / np.sqrt(var + eps) * np.asarray(gamma).astype(f64) + np.asarray(beta).astype(f64))


def group_norm_nhwc_exact(x, gamma, beta, groups, eps, silu_act=False):
    """GroupNorm on NHWC in exact arithmetic with ONE final rounding (what the fused device kernel computes)."""
    n, h, w, c = x.shape
    x64 = np.asarray(x).astype(f64).reshape(n, h * w, groups, c // groups)
    mean = x64.mean(axis=(1, 3), keepdims=True)
    var = ((x64 - mean) ** 2).mean(axis=(1, 3), keepdims=True)
    y = ((x64 - mean) / np.sqrt(var + eps)).reshape(n, h, w, c)
    y = y * np.asarray(gamma).astype(f64).reshape(c) + np.asarray(beta).astype(f64).reshape(c)
    if silu_act:
        y = y / (1.0 + np.exp(-y))
    return r16(y)


def group_norm_decomposed_nchw(x_nchw, gamma, beta, groups, eps):
    """The exported GroupNorm chain on NCHW data: Reshape[1,G,-1] -> InstanceNorm(1,0) -> Reshape -> Mul -> Add."""
    n, c, h, w = x_nchw.shape
    assert n == 1
    r = np.asarray(x_nchw).reshape(1, groups, -1)
    i = instance_norm(r, np.ones(groups, f32), np.zeros(groups, f32), eps).reshape(n, c, h, w)
    return binary("add", binary("mul", i, np.asarray(gamma).reshape(c, 1, 1)), np.asarray(beta).reshape(c, 1, 1))


# ---- attention -----------------------------------------------------------------------------------------------------------
def attention_fused_ops(q, k_dt, v, scale, parts=2):
    """AttentionFusedOps (:6696-6929): per head, per Q row-chunk: S=f16(Q K^T); S=f16(S*s); P=softmax; O=f16(P V).
    q:[h,Tq,d], k_dt:[h,d,Tkv] (already transposed), v:[h,Tkv,d], scale: f16 scalar or None."""
    h, tq, d = q.shape
    while tq % parts:
        parts += 1
    rows = tq // parts
    out = np.empty((h, tq, d), f16)
    for i in range(h):
        for j in range(parts):
            qs = q[i, j * rows:(j + 1) * rows]
            s = matmul(qs, k_dt[i])
            if scale is not None:
                s = binary("mul", s, np.asarray(scale, f16).reshape(1))
            p = softmax_last(s)
            out[i, j * rows:(j + 1) * rows] = matmul(p, v[i])
    return out


def attention_exact(q, k, v, scale):
    """softmax(scale * Q K^T) V in double, one final rounding. q:[h,Tq,d], k:[h,Tkv,d], v:[h,Tkv,d]."""
    s = np.einsum("hqd,hkd->hqk", q.astype(f64), k.astype(f64)) * float(scale)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return r16(np.einsum("hqk,hkd->hqd", p, v.astype(f64)))


# ---- quantisation (integer path: bit-exact contracts) -----------------------------------------------------
def range_to_scale(lo, hi):
    """Model::range_to_scale (:3234): range forced to include 0; scale = (float)((hi - lo) / 255.0) where hi - lo is a FLOAT subtraction
    (rounded) and only the division runs in double; zp = (uint8_t)(|lo| / scale), a float division, truncated.  (Pinned against the
    reference's push_tensor quantisation in tests/test_qu8_oracle.py: the double subtraction differs by an ulp of the scale on some ranges.)"""
    lo, hi = min(float(lo), 0.0), max(float(hi), 0.0)
    d = np.float32(np.float32(hi) - np.float32(lo))
    scale = np.float32(np.float64(d) / 255.0)
    zp = int(np.uint8(int(np.float32(abs(np.float32(lo))) / scale)))
    return scale, zp


def quantize_u8(x, scale, zp):
    """f32 -> u8 (XnnPack::convert_qu8 :802 -> xnn f32->qu8 convert): clamp(rne(x * (1.0f/scale)) + zp, 0, 255)."""
    inv = np.float32(1.0) / np.float32(scale)
    r = np.rint(np.asarray(x, f32) * inv) + np.float32(zp)
    return np.clip(r, 0, 255).astype(np.uint8)


def dequantize_u8(q, scale, zp, dtype=f32):
    """u8 -> f32: (float)((int)q - zp) * scale ; to f16 through the f32 staging value (:3353-3434)."""
    v = (np.asarray(q).astype(np.int32) - int(zp)).astype(f32) * np.float32(scale)
    return v.astype(dtype)
