"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's uint8-arithmetic ops (m_use_uint8_arithmetic), for the round that
builds the W8A8 VAE path.  Every function cites the reference lines it follows; tools/qu8_probe.py pins them against the reference's own
intermediates (oracle/_ref) on the miniature qu8 VAE.  A uint8 tensor is (codes, scale: float32, zero_point: int)."""
from __future__ import annotations

import numpy as np

from .np_ops import dequantize_u8, quantize_u8, range_to_scale  # noqa: F401  (shared integer contracts)

f32, f64 = np.float32, np.float64


def sigmoid_u8(q, s_in, z_in, s_out, z_out):
    """Sigmoid, uint8 branch (onnxstream.cpp:4412-4481): dequantise, 1 / (1 + exp(-x)) in fp32 (std::exp on float), requantise."""
    x = dequantize_u8(q, s_in, z_in)
    y = (f32(1) / (f32(1) + np.exp(-x, dtype=f32))).astype(f32)
    return quantize_u8(y, s_out, z_out)


def instance_norm_u8(q, s_in, z_in, scale, bias, eps, s_out, z_out):
    """InstanceNormalization, uint8 branch (onnxstream.cpp:4987-5043) on [1, C, L]: per channel mean = double sum of the dequantised floats
    / L; variance = double sum of (float)(x - mean)^2 computed in float / L; y = scale * (x - mean) / sqrt(var + eps) + b evaluated in
    double and rounded to float; requantise."""
    x = dequantize_u8(q, s_in, z_in)                       # [1, C, L] float32
    mean = x.astype(f64).sum(-1, keepdims=True) / x.shape[-1]
    dev = (x.astype(f64) - mean).astype(f32)              # float dev = buffer[k] - mean
    var = (dev * dev).astype(f32).astype(f64).sum(-1, keepdims=True) / x.shape[-1]
    sr = np.sqrt(var + f64(f32(eps)))
    sc = np.asarray(scale, f32).reshape(1, -1, 1).astype(f64)
    b = np.asarray(bias, f32).reshape(1, -1, 1).astype(f64)
    y = (sc * (x.astype(f64) - mean) / sr + b).astype(f32)
    return quantize_u8(y, s_out, z_out)


def conv_bias_i32(bias_f32, s_x, s_w):
    """Conv, uint8 branch: the fp32 bias is rescaled IN PLACE to int32 by (int32_t)(b / (x_scale * w_scale)) (onnxstream.cpp:4639-4660)."""
    sc = f32(s_x) * f32(s_w)
    return np.trunc(np.asarray(bias_f32, f32) / sc).astype(np.int32)


def requant_fp32(acc_i32, scale, z_out):
    """XNNPACK qu8 fp32 requantisation (gemm/igemm minmax fp32 microkernels): float(acc) * scale, clamp to [0 - z, 255 - z] as float,
    round to nearest even, + zero point."""
    v = acc_i32.astype(f32) * f32(scale)
    v = np.minimum(np.maximum(v, f32(0 - z_out)), f32(255 - z_out))
    return (np.rint(v).astype(np.int32) + int(z_out)).astype(np.uint8)


def conv2d_nhwc_u8(q, s_x, z_x, w_ohwi, s_w, z_w, bias_f32, pads, strides, s_out, z_out):
    """Conv, uint8 branch -> XnnPack::convolution<uint8_t,int32_t> (onnxstream.cpp:1292, :1458-1491): acc = sum (x - zx)(w - zw) + bias_i32
    (padding contributes zero: XNNPACK pads with the input zero point), fp32 requantisation with scale = sx * sw / sy."""
    n, H, W, C = q.shape
    O, KH, KW, _ = w_ohwi.shape
    pt, pl, pb, pr = pads
    x = np.full((n, H + pt + pb, W + pl + pr, C), int(z_x), np.int32)
    x[:, pt:pt + H, pl:pl + W, :] = q.astype(np.int32)
    x -= int(z_x)
    w = w_ohwi.astype(np.int32) - int(z_w)
    Ho, Wo = (H + pt + pb - KH) // strides[0] + 1, (W + pl + pr - KW) // strides[1] + 1
    acc = np.zeros((n, Ho, Wo, O), np.int64)
    for kh in range(KH):
        for kw in range(KW):
            patch = x[:, kh:kh + (Ho - 1) * strides[0] + 1:strides[0], kw:kw + (Wo - 1) * strides[1] + 1:strides[1], :]
            acc += np.einsum("nhwc,oc->nhwo", patch.astype(np.int64), w[:, kh, kw, :].astype(np.int64))
    if bias_f32 is not None:
        acc += conv_bias_i32(bias_f32, s_x, s_w).astype(np.int64)
    scale = f32(f32(s_x) * f32(s_w)) / f32(s_out)
    return requant_fp32(acc.astype(np.int32), scale, z_out)


def mul_u8(a, s_a, z_a, b, s_b, z_b, s_out, z_out):
    """Mul, uint8 branch -> XnnPack::multiply<uint8_t> -> xnn_run_binary_elementwise_nd(multiply, quint8) (onnxstream.cpp:846-927, :3977-3996):
    XNNPACK's qu8 vmul fp32 microkernel -- (a - za)(b - zb) as int32, * (sa * sb / so) in fp32, clamp, round to nearest even, + zo."""
    acc = (a.astype(np.int32) - int(z_a)) * (b.astype(np.int32) - int(z_b))
    scale = f32(f32(s_a) * f32(s_b)) / f32(s_out)
    return requant_fp32(acc, scale, z_out)


def add_u8(a, s_a, z_a, b, s_b, z_b, s_out, z_out):
    """Add, uint8 branch -> XnnPack::add<uint8_t> -> xnn_run_binary_elementwise_nd(add, quint8) (onnxstream.cpp:1666, :5105-5124): XNNPACK's
    qu8 vadd fixed-point microkernel -- the two input/output scale ratios become integer multipliers with 20 bits for the larger one
    (shift = 20 - exponent of the larger ratio), acc = bias + a * am + b * bm, arithmetic shift right with the rounding folded into bias."""
    a_os, b_os = f32(s_a) / f32(s_out), f32(s_b) / f32(s_out)
    mx = max(a_os, b_os)
    exponent = int((np.asarray(mx, f32).view(np.uint32) >> 23)) - 127
    shift = 20 - exponent
    am = int(np.rint(f32(a_os) * f32(2.0 ** shift)))
    bm = int(np.rint(f32(b_os) * f32(2.0 ** shift)))
    bias = (1 << (shift - 1)) - am * int(z_a) - bm * int(z_b)
    acc = bias + a.astype(np.int64) * am + b.astype(np.int64) * bm
    out = (acc >> shift) + int(z_out)
    return np.clip(out, 0, 255).astype(np.uint8)


def matmul_u8(a, s_a, z_a, b, s_b, z_b, s_out, z_out):
    """MatMul, uint8 branch -> XnnPack::matrix_multiply<uint8_t, void> per batch item (onnxstream.cpp:5779-5825, :1035-1215): XNNPACK qu8
    fully-connected -- acc = sum_k (a - za)(b - zb) in int32 (no bias), fp32 requantisation with scale = sa * sb / so.
    a:[..., M, K], b:[..., K, N] (or 2-D [K, N] broadcast)."""
    acc = np.matmul(a.astype(np.int64) - int(z_a), b.astype(np.int64) - int(z_b))
    scale = f32(f32(s_a) * f32(s_b)) / f32(s_out)
    return requant_fp32(acc.astype(np.int32), scale, z_out)


def softmax_u8(q, s_in, axis=-1):
    """Softmax, uint8 branch -> XnnPack::softmax<uint8_t> -> XNNPACK qu8 softmax (onnxstream.cpp:1958-2060): output scale 1/256, zero point 0;
    lookup table t[i] = lrint(min(UINT32_MAX / channels, 2^23 - 1) * exp((i - 255) * s_in)); per row y = min(255, ((t[x + 255 - max] << 8)
    + (sum >> 1)) / sum).  (In the oracle the table operator itself is the shim's restatement of the same algorithm.)"""
    x = np.moveaxis(np.asarray(q), axis, -1)
    c = x.shape[-1]
    qscale = min(float(np.iinfo(np.uint32).max) / c, 8388607.0)
    t = np.rint(qscale * np.exp((np.arange(256, dtype=np.float64) - 255.0) * float(f32(s_in)))).astype(np.uint64)
    m = x.max(-1, keepdims=True).astype(np.int64)
    tv = t[(x.astype(np.int64) + 255 - m)]
    s = tv.sum(-1, keepdims=True)
    y = np.minimum(((tv << np.uint64(8)) + (s >> np.uint64(1))) // s, 255).astype(np.uint8)
    return np.moveaxis(y, -1, axis), f32(1.0 / 256.0), 0


def resize_nearest_u8(q, scale_h, scale_w):
    """Resize (nearest, asymmetric, floor), NCHW: the codes are copied, scale and zero point carried over (onnxstream.cpp:6120-6315)."""
    n, c, h, w = q.shape
    hi = np.minimum((np.arange(int(h * scale_h)) / scale_h).astype(np.int64), h - 1)
    wi = np.minimum((np.arange(int(w * scale_w)) / scale_w).astype(np.int64), w - 1)
    return q[:, :, hi][:, :, :, wi]


def percentiles(x, from_left=0.001, from_right=0.001, threads=1, chunk=16384):
    """Model::get_percentiles on fp32 data (onnxstream.cpp:3104-3231, FloatAsUInt::get_percentiles :2300-2386): the tensor is split evenly
    over `threads` workers (get_start_and_end :3091), each worker walks its span in chunks of 16 K elements (the 64 KiB per-thread buffer),
    sorts a chunk and takes the element (size_t)(n * from_left) from the bottom and (size_t)(n * from_right) from the top of its FINITE
    values; the result is the min of the lows and the max of the highs over all chunks -- so it depends on the thread count and the chunk
    size, exactly like the reference's.  Returns (lo, hi) or None."""
    flat = np.asarray(x, f32).ravel()
    size = flat.size
    per = max(size // threads, 1)
    lo, hi, found = np.inf, -np.inf, False
    for i in range(threads):
        start = i * per
        end = size if i >= threads - 1 else (i + 1) * per
        if start >= end or start >= size or end > size:
            continue
        for j in range(start, end, chunk):
            c = np.sort(flat[j:min(end, j + chunk)])
            n = c.size
            c = c[np.isfinite(c)]
            kl, kr = int(f32(n) * f32(from_left)), int(f32(n) * f32(from_right))
            if kl >= c.size or kr >= c.size:
                continue
            lo, hi, found = min(lo, float(c[kl])), max(hi, float(c[c.size - 1 - kr])), True
    if not found or not np.isfinite(lo) or not np.isfinite(hi) or lo >= hi:
        return None
    return f32(lo), f32(hi)


def quantize_dynamic(x, threads=1):
    """Model::quantize (onnxstream.cpp:3247-3352) = what push_tensor does to a pushed fp32 input under uint8 arithmetic (:3024-3028): 0.1 %
    percentiles from each end -> range_to_scale -> f32 -> u8.  Returns (codes, scale, zero_point)."""
    lo, hi = percentiles(x, 0.001, 0.001, threads)
    scale, zp = range_to_scale(lo, hi)
    return quantize_u8(x, scale, zp), scale, zp
