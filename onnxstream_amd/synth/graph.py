"""Emitter for OnnxStream ``model.txt`` graphs + raw ``.bin`` weight files (numpy only).

On-disk conventions follow the reference runtime's parser (reference src/onnxstream.cpp:2445-2616) and its
exporter's habits (SURVEY.md section 5 "On-disk formats"):

* one op per line: ``name:Type*input:T;T*output:T[*attr:val;attr:val]``
* activation tensor ``T`` = ``name(d0,d1,..)``; weight ``T`` = ``file.bin(dtype:d0,d1,..)`` with dtype in
  ``float32|float16|int64|uint8[scale,zero_point]``; a scalar has an empty shape (``file.bin(float16:)``)
* Conv weights are referenced as ``<w>_nchw.bin`` with the OIHW shape, while the file the runtime actually loads is
  ``<w>_nhwc.bin`` holding OHWI data (onnxstream.cpp:2666-2692) -- only the ``_nhwc`` file is written here.
* MatMul / Gemm matrices are stored ``[K,N]`` row-major.

Weights go to a *sink*: a directory (``DirSink``) or an in-memory dict (``MemSink``) that a RAM weights provider
is filled from without touching the disk (used by bench.py on the GPU box).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

_DT = {"float32": np.float32, "float16": np.float16, "int64": np.int64, "uint8": np.uint8}


def mangle(name: str) -> str:
    """Exporter-style name mangling: every non-alphanumeric char -> ``_HEX_`` (reference src/bindings.py:310)."""
    return "".join(c if c.isalnum() else f"_{ord(c):X}_" for c in name)


class MemSink:
    def __init__(self):
        self.files: Dict[str, np.ndarray] = {}

    def write(self, fname: str, arr: np.ndarray):
        self.files[fname] = np.ascontiguousarray(arr)

    def write_text(self, fname: str, text: str):
        self.files[fname] = text


class DirSink:
    def __init__(self, path: str):
        self.path = path
        os.makedirs(path, exist_ok=True)

    def write(self, fname: str, arr: np.ndarray):
        np.ascontiguousarray(arr).tofile(os.path.join(self.path, fname))

    def write_text(self, fname: str, text: str):
        with open(os.path.join(self.path, fname), "w") as f:
            f.write(text)


class T:
    """A graph value: activation (name + logical shape) or weight file (token + shape)."""

    __slots__ = ("name", "shape", "token", "is_weight")

    def __init__(self, name, shape, token=None, is_weight=False):
        self.name = name
        self.shape = tuple(int(s) for s in shape)
        self.token = token if token is not None else f"{name}({','.join(str(s) for s in self.shape)})"
        self.is_weight = is_weight


def quantize_u8(w: np.ndarray) -> Tuple[np.ndarray, float, int]:
    """uint8 weight quantisation as the reference exporter does it (onnx2txt.ipynb ``quantize``): 0.1 % percentiles
    from each end, range forced to include 0, scale = range/255, zero_point = int(|lo|/scale) clipped to 255."""
    flat = np.sort(w.astype(np.float32).ravel())
    n = flat.size
    k = int(n * 0.001)
    lo, hi = float(flat[k]), float(flat[n - 1 - k])
    lo, hi = min(lo, 0.0), max(hi, 0.0)
    if hi == lo:
        hi = lo + 1.0
    scale = (hi - lo) / 255.0
    zp = min(int(abs(lo) / scale), 255)
    q = np.clip(np.rint(w.astype(np.float32) / np.float32(scale)) + zp, 0, 255).astype(np.uint8)
    return q, float(np.float32(scale)), zp


class GraphBuilder:
    def __init__(self, sink, wdtype: str = "float16", seed: int = 1234, quant_weights: bool = False, quant_all: bool = False):
        self.sink = sink
        self.wdtype = wdtype
        self.quant = quant_weights or quant_all
        # quant_all: the reference exporter's full uint8 model (onnx2txt.ipynb ``quantize``: EVERY float initializer except Conv biases,
        # InstanceNormalization scale/bias and Resize scales) -- what ``m_use_uint8_arithmetic`` needs (the vae_decoder_qu8 directory)
        self.quant_all = quant_all
        self.rng = np.random.default_rng(seed)
        self.lines: List[str] = []
        self._uid = 0
        self._wnames = set()
        self.inputs: List[T] = []
        self.n_params = 0

    # -- values ------------------------------------------------------------------------------------
    def input(self, name: str, shape) -> T:
        t = T(mangle(name), shape)
        self.inputs.append(t)
        return t

    def _fresh(self, base: str) -> str:
        self._uid += 1
        return f"{mangle(base)}_{self._uid}"

    def weight(self, name: str, arr: np.ndarray, dtype: Optional[str] = None, conv: bool = False,
               allow_quant: bool = True, q8_exempt: bool = False) -> T:
        """Write ``arr`` (given in the ONNX initializer layout: OIHW for conv, [K,N] for matmul) and return its token."""
        dtype = dtype or self.wdtype
        if self.quant_all and q8_exempt and dtype == "float16":
            dtype = "float32"      # the exporter leaves what it does not quantise in the ONNX file's own fp32 (the qu8 Conv wants an fp32 bias)
        base = mangle(name)
        assert base not in self._wnames, base
        self._wnames.add(base)
        shape = arr.shape
        self.n_params += int(arr.size) if dtype != "int64" else 0
        data = arr
        if conv:
            data = np.transpose(arr, (0, 2, 3, 1))  # OIHW -> OHWI on disk
            fname_ref, fname_disk = base + "_nchw.bin", base + "_nhwc.bin"
        else:
            fname_ref = fname_disk = base + ".bin"
        if dtype in ("float16", "float32") and ((self.quant and allow_quant and arr.size >= 1024) or (self.quant_all and not q8_exempt)):
            q, scale, zp = quantize_u8(data)
            self.sink.write(fname_disk, q)
            tystr = f"uint8[{scale!r},{zp}]"
        else:
            self.sink.write(fname_disk, data.astype(_DT[dtype]))
            tystr = dtype
        token = f"{fname_ref}({tystr}:{','.join(str(s) for s in shape)})"
        return T(fname_ref, shape, token, True)

    def const_i64(self, name: str, vals: Sequence[int]) -> T:
        return self.weight(name, np.asarray(vals, dtype=np.int64), dtype="int64")

    def randn(self, shape, std: float) -> np.ndarray:
        return (self.rng.standard_normal(size=shape, dtype=np.float32) * np.float32(std))

    # -- ops ---------------------------------------------------------------------------------------
    def op(self, name: str, typ: str, inputs: Iterable[Optional[T]], out_shapes, attrs: Optional[dict] = None,
           out_names: Optional[Sequence[str]] = None):
        single = out_shapes and not isinstance(out_shapes[0], (tuple, list))
        shapes = [out_shapes] if single else list(out_shapes)
        outs = []
        for i, s in enumerate(shapes):
            nm = out_names[i] if out_names else self._fresh(f"{name}_out{i}")
            outs.append(T(mangle(nm) if out_names else nm, s))
        ins = ";".join("" if t is None else t.token for t in inputs)
        line = f"{name}:{typ}*input:{ins}*output:{';'.join(o.token for o in outs)}"
        if attrs:
            line += "*" + ";".join(f"{k}:{v}" for k, v in attrs.items())
        self.lines.append(line)
        return outs[0] if single else outs

    def text(self) -> str:
        return "\n".join(self.lines) + "\n"

    def finish(self) -> str:
        txt = self.text()
        self.sink.write_text("model.txt", txt)
        return txt

    # -- layer helpers (shapes are logical ONNX/NCHW shapes) ------------------------------------------
    def conv(self, name, x: T, cout: int, k: int = 3, stride: int = 1, pad: Optional[int] = None, bias: bool = True,
             std: Optional[float] = None) -> T:
        n, cin, h, w = x.shape
        pad = (k // 2) if pad is None else pad
        std = std if std is not None else (1.0 / np.sqrt(cin * k * k))
        wt = self.weight(f"{name}.weight", self.randn((cout, cin, k, k), std), conv=True)
        ins = [x, wt]
        if bias:
            ins.append(self.weight(f"{name}.bias", self.randn((cout,), 0.02), allow_quant=False, q8_exempt=True))
        ho = (h + 2 * pad - k) // stride + 1
        wo = (w + 2 * pad - k) // stride + 1
        return self.op(name, "Conv", ins, (n, cout, ho, wo),
                       {"dilations": "1,1", "group": "1", "kernel_shape": f"{k},{k}",
                        "pads": f"{pad},{pad},{pad},{pad}", "strides": f"{stride},{stride}"})

    def conv1d(self, name, x: T, cout: int, k: int = 3, stride: int = 1, pad: Optional[int] = None, bias: bool = True) -> T:
        """Conv over [N, C, L] as the ONNX exporter writes it: one-element dilations / kernel_shape / strides, two pads (the reference lifts it to 2-D,
        src/onnxstream.cpp:4521-4544)."""
        n, cin, length = x.shape
        pad = (k // 2) if pad is None else pad
        wt = self.weight(f"{name}.weight", self.randn((cout, cin, k, 1), 1.0 / np.sqrt(cin * k)), conv=True)   # (model.txt names a Conv1D's filter bank as [O, I, k, 1]: get_tensor_data :2681 wants four dims)
        ins = [x, wt]
        if bias:
            ins.append(self.weight(f"{name}.bias", self.randn((cout,), 0.02), allow_quant=False, q8_exempt=True))
        lo = (length + 2 * pad - k) // stride + 1
        return self.op(name, "Conv", ins, (n, cout, lo), {"dilations": "1", "group": "1", "kernel_shape": f"{k}", "pads": f"{pad},{pad}", "strides": f"{stride}"})

    def matmul_w(self, name, x: T, n_out: int, std: Optional[float] = None) -> T:
        k = x.shape[-1]
        std = std if std is not None else (1.0 / np.sqrt(k))
        wt = self.weight(f"{name}.weight", self.randn((k, n_out), std))
        return self.op(name, "MatMul", [x, wt], x.shape[:-1] + (n_out,))

    def add_bias(self, name, x: T, std: float = 0.02) -> T:
        b = self.weight(f"{name}.bias", self.randn((x.shape[-1],), std), allow_quant=False)
        return self.op(name, "Add", [x, b], x.shape)

    def linear(self, name, x: T, n_out: int, bias: bool = True) -> T:
        y = self.matmul_w(name + "/MatMul", x, n_out)
        return self.add_bias(name + "/Add", y) if bias else y

    def gemm(self, name, x: T, n_out: int) -> T:
        k = x.shape[-1]
        wt = self.weight(f"{name}.weight", self.randn((k, n_out), 1.0 / np.sqrt(k)))
        b = self.weight(f"{name}.bias", self.randn((n_out,), 0.02), allow_quant=False)
        return self.op(name, "Gemm", [x, wt, b], (x.shape[0], n_out))

    def binary(self, name, typ, a: T, b: T) -> T:
        shape = tuple(np.broadcast_shapes(a.shape, b.shape))
        return self.op(name, typ, [a, b], shape)

    def unary(self, name, typ, x: T) -> T:
        return self.op(name, typ, [x], x.shape)

    def silu(self, name, x: T) -> T:
        s = self.unary(name + "/Sigmoid", "Sigmoid", x)
        return self.binary(name + "/Mul", "Mul", x, s)

    def reshape(self, name, x: T, shape) -> T:
        shape = tuple(int(s) for s in shape)
        assert int(np.prod(shape)) == int(np.prod(x.shape)), (name, x.shape, shape)
        c = self.const_i64(f"{name}.shape", shape)
        return self.op(name, "Reshape", [x, c], shape, {"allowzero": "0"})

    def transpose(self, name, x: T, perm) -> T:
        return self.op(name, "Transpose", [x], tuple(x.shape[p] for p in perm), {"perm": ",".join(map(str, perm))})

    def unsqueeze(self, name, x: T, axis: int) -> T:
        c = self.const_i64(f"{name}.axes", [axis])
        shape = list(x.shape)
        shape.insert(axis if axis >= 0 else len(shape) + 1 + axis, 1)
        return self.op(name, "Unsqueeze", [x, c], tuple(shape))

    def concat(self, name, xs: Sequence[T], axis: int) -> T:
        shape = list(xs[0].shape)
        shape[axis] = sum(x.shape[axis] for x in xs)
        return self.op(name, "Concat", list(xs), tuple(shape), {"axis": str(axis)})

    def slice_last(self, name, x: T, start: int, end: int) -> T:
        s = self.const_i64(f"{name}.starts", [start])
        e = self.const_i64(f"{name}.ends", [end])
        a = self.const_i64(f"{name}.axes", [-1])
        st = self.const_i64(f"{name}.steps", [1])
        return self.op(name, "Slice", [x, s, e, a, st], x.shape[:-1] + (end - start,))

    def scalar(self, name, val: float, dtype: Optional[str] = None) -> T:
        return self.weight(name, np.asarray(val, dtype=np.float32).reshape(()), dtype=dtype, allow_quant=False)

    def group_norm(self, name, x: T, groups: int = 32, eps: float = 1e-5) -> T:
        n, c, h, w = x.shape
        r = self.reshape(name + "/Reshape", x, (1, groups, c * h * w // groups))
        ones = self.weight(f"{name}.in_scale", np.ones((groups,), np.float32), allow_quant=False, q8_exempt=True)
        zeros = self.weight(f"{name}.in_bias", np.zeros((groups,), np.float32), allow_quant=False, q8_exempt=True)
        i = self.op(name + "/InstanceNormalization", "InstanceNormalization", [r, ones, zeros], r.shape,
                    {"epsilon": repr(float(eps))})
        r2 = self.reshape(name + "/Reshape_1", i, (n, c, h, w))
        g = self.weight(f"{name}.weight", 1.0 + self.randn((c, 1, 1), 0.1), allow_quant=False)
        b = self.weight(f"{name}.bias", self.randn((c, 1, 1), 0.1), allow_quant=False)
        m = self.binary(name + "/Mul", "Mul", r2, g)
        return self.binary(name + "/Add", "Add", m, b)

    def layer_norm(self, name, x: T, eps: float = 1e-5) -> T:
        c = x.shape[-1]
        red = x.shape[:-1] + (1,)
        mean = self.op(name + "/ReduceMean", "ReduceMean", [x], red, {"axes": "-1", "keepdims": "1"})
        sub = self.binary(name + "/Sub", "Sub", x, mean)
        p = self.binary(name + "/Pow", "Pow", sub, self.scalar(f"{name}.pow_exp", 2.0))
        var = self.op(name + "/ReduceMean_1", "ReduceMean", [p], red, {"axes": "-1", "keepdims": "1"})
        ve = self.binary(name + "/Add", "Add", var, self.scalar(f"{name}.eps", eps))
        sd = self.unary(name + "/Sqrt", "Sqrt", ve)
        d = self.binary(name + "/Div", "Div", sub, sd)
        g = self.weight(f"{name}.weight", 1.0 + self.randn((c,), 0.1), allow_quant=False)
        b = self.weight(f"{name}.bias", self.randn((c,), 0.1), allow_quant=False)
        m = self.binary(name + "/Mul", "Mul", d, g)
        return self.binary(name + "/Add_1", "Add", m, b)
