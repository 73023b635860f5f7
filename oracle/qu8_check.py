"""TEST INFRASTRUCTURE ONLY -- pins oracle/np_qu8.py (the numpy restatement of the reference's uint8-arithmetic ops) against the reference
itself: runs oracle/_ref with m_use_uint8_arithmetic keeping EVERY op output raw (codes + scale + zero point), then re-derives each op's
output from its inputs with the restatements and counts differing codes per op type.  Used by tests/test_qu8_oracle.py and
tools/qu8_probe.py."""
import collections
import ctypes
import os
import tempfile

import numpy as np

from onnxstream_amd.bindings import Model
from . import np_qu8 as Q
from . import ref as oref


def parse_model(path):
    ops = []
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            continue
        head, rest = line.split("*input:", 1)
        name, typ = head.rsplit(":", 1)
        ins, rest = rest.split("*output:", 1)
        outs, _, attrs = rest.partition("*")
        def toks(s):
            res, depth, cur = [], 0, ""
            for ch in s:
                if ch == "(":
                    depth += 1
                if ch == ")":
                    depth -= 1
                if ch == ";" and depth == 0:
                    res.append(cur); cur = ""
                else:
                    cur += ch
            if cur:
                res.append(cur)
            return res
        ops.append(dict(name=name, type=typ, inputs=toks(ins), outputs=toks(outs), attrs=dict(a.split(":", 1) for a in attrs.split(";") if ":" in a)))
    return ops


def tname(tok):
    return tok.split("(", 1)[0]


def run_u8_all(model_dir, inputs, ranges):
    m = Model(oref.REF_LIB, 1, "ram+nocache")
    oref._extra(m.lib)
    lib = m.lib
    lib.ref_get_tensor_any.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t),
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_void_p)]
    lib.ref_get_tensor_any.restype = ctypes.c_size_t
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "r.txt")
        open(fn, "w", newline="").write(ranges)
        assert not lib.ref_read_range_data(m.handle, fn.encode())
    m._set_option("use_uint8_arithmetic", 1)
    m.read_file(os.path.join(model_dir, "model.txt"))
    ops = parse_model(os.path.join(model_dir, "model.txt"))
    for op in ops:
        for o in op["outputs"]:
            lib.model_add_extra_output(m.handle, tname(o).encode())
    # a non-empty convert set = ONLY its members are dequantised / transposed back at the end of run(): everything else stays raw
    lib.ref_add_outputs_convert_exclusion.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    lib.ref_add_outputs_convert_exclusion(m.handle, b"__none__")
    for k, v in inputs.items():
        a = np.ascontiguousarray(v, np.float32)
        dims = (ctypes.c_uint * a.ndim)(*a.shape)
        assert not lib.ref_push_tensor_f32(m.handle, m._name(k), a.ndim, dims, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    vals = {}

    def grab(nm):
        dt, rank, scale, zp, ptr = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_float(), ctypes.c_int(), ctypes.c_void_p()
        shape = (ctypes.c_size_t * 8)()
        n = lib.ref_get_tensor_any(m.handle, nm.encode(), ctypes.byref(dt), ctypes.byref(rank), shape, ctypes.byref(scale), ctypes.byref(zp), ctypes.byref(ptr))
        if not n:
            return None
        npdt = {1: np.uint8, 2: np.uint16, 3: np.float32, 4: np.int64}[dt.value]
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(npdt))), shape=(n,)).copy()
        return dict(data=arr.reshape([shape[i] for i in range(rank.value)]), scale=np.float32(scale.value), zp=int(zp.value), dtype=dt.value)

    for k in inputs:          # the pushed inputs, as push_tensor quantised them (they are consumed by the pass)
        nm = m._name(k).decode()
        v = grab(nm)
        if v is not None:
            vals[nm] = v
    m.run()
    for op in ops:
        for o in op["outputs"] + op["inputs"]:
            nm = tname(o)
            if nm in vals or nm.endswith(".bin"):
                continue
            dt, rank, scale, zp, ptr = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_float(), ctypes.c_int(), ctypes.c_void_p()
            shape = (ctypes.c_size_t * 8)()
            n = lib.ref_get_tensor_any(m.handle, nm.encode(), ctypes.byref(dt), ctypes.byref(rank), shape, ctypes.byref(scale), ctypes.byref(zp), ctypes.byref(ptr))
            if not n:
                continue
            npdt = {1: np.uint8, 2: np.uint16, 3: np.float32, 4: np.int64}[dt.value]
            arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(npdt))), shape=(n,)).copy()
            vals[nm] = dict(data=arr.reshape([shape[i] for i in range(rank.value)]), scale=np.float32(scale.value), zp=int(zp.value), dtype=dt.value)
    m.close()
    return ops, vals



def weights_of(model_dir, tok):
    """(array, scale, zp) of a constant token `file(dtype[scale,zp]:shape)` read from disk (conv weights: the OHWI file)."""
    name, rest = tok.split("(", 1)
    ty, shp = rest[:-1].rsplit(":", 1)
    shape = [int(v) for v in shp.split(",")] if shp else []
    fn = name.replace("_nchw.bin", "_nhwc.bin")
    if ty.startswith("uint8"):
        sc, zp = ty[6:-1].split(",")
        a = np.fromfile(model_dir + fn, np.uint8)
        if name.endswith("_nchw.bin"):
            a = a.reshape(shape[0], shape[2], shape[3], shape[1])
        else:
            a = a.reshape(shape)
        return a, np.float32(float(sc)), int(zp)
    a = np.fromfile(model_dir + fn, {"float32": np.float32, "float16": np.float16, "int64": np.int64}[ty]).reshape(shape)
    return a, None, None




def verify(d, inputs, ranges):
    """-> {op type: [ops checked, codes compared, codes that differ]} for the uint8 model in directory d."""
    ops, vals = run_u8_all(d, inputs, ranges)
    stats = collections.defaultdict(lambda: [0, 0, 0])      # ops checked, elements, mismatches
    prod = {tname(o): op["type"] for op in ops for o in op["outputs"]}

    def nhwc(tok):
        """the tensor as the convolution sees it: Conv outputs are stored NHWC, everything else logical NCHW (transposed on fetch, :2914-2955)"""
        v = vals[tname(tok)]
        return v["data"] if prod.get(tname(tok)) == "Conv" else np.ascontiguousarray(v["data"].transpose(0, 2, 3, 1))
    for op in ops:
        t = op["type"]
        out = vals.get(tname(op["outputs"][0]))
        ins = [vals.get(tname(i)) for i in op["inputs"]]
        if out is None:
            continue
        got = None
        if t == "Sigmoid" and ins[0] is not None:
            got = Q.sigmoid_u8(ins[0]["data"], ins[0]["scale"], ins[0]["zp"], out["scale"], out["zp"])
        elif t == "InstanceNormalization" and ins[0] is not None:
            sc = weights_of(d, op["inputs"][1])[0]
            b = weights_of(d, op["inputs"][2])[0]
            got = Q.instance_norm_u8(ins[0]["data"], ins[0]["scale"], ins[0]["zp"], sc, b, float(op["attrs"]["epsilon"]), out["scale"], out["zp"])
        elif t == "Conv" and ins[0] is not None:
            w, sw, zw = weights_of(d, op["inputs"][1])
            bias = weights_of(d, op["inputs"][2])[0] if len(op["inputs"]) > 2 else None
            pads = [int(v) for v in op["attrs"]["pads"].split(",")]
            st = [int(v) for v in op["attrs"]["strides"].split(",")]
            got = Q.conv2d_nhwc_u8(nhwc(op["inputs"][0]), ins[0]["scale"], ins[0]["zp"], w, sw, zw, bias, (pads[0], pads[1], pads[2], pads[3]), st, out["scale"], out["zp"])
        elif t in ("Mul", "Add"):
            def operand(tok):
                nm = tname(tok)
                if nm in vals:
                    v = vals[nm]
                    dta = v["data"]
                    if prod.get(nm) == "Conv":       # stored NHWC -> the binary ops see the logical NCHW tensor
                        dta = dta.transpose(0, 3, 1, 2)
                    return dta, v["scale"], v["zp"]
                return weights_of(d, tok)
            (a, sa, za), (b2, sb, zb) = operand(op["inputs"][0]), operand(op["inputs"][1])
            if sa is not None and sb is not None:
                fn = Q.mul_u8 if t == "Mul" else Q.add_u8
                got = fn(a, sa, za, b2, sb, zb, out["scale"], out["zp"])
                if got.shape != out["data"].shape:
                    got = np.broadcast_to(got, out["data"].shape)
        elif t == "MatMul":
            def operand2(tok):
                nm = tname(tok)
                if nm in vals:
                    v = vals[nm]
                    return v["data"], v["scale"], v["zp"]
                return weights_of(d, tok)
            (a, sa, za), (b2, sb, zb) = operand2(op["inputs"][0]), operand2(op["inputs"][1])
            got = Q.matmul_u8(a, sa, za, b2, sb, zb, out["scale"], out["zp"])
        elif t == "Softmax" and ins[0] is not None:
            got, so_, zo_ = Q.softmax_u8(ins[0]["data"], ins[0]["scale"], int(op["attrs"].get("axis", "-1")))
            assert abs(float(out["scale"]) - float(so_)) < 1e-12 and out["zp"] == zo_, (out["scale"], out["zp"])
        elif t == "Resize" and ins[0] is not None:
            got = Q.resize_nearest_u8(ins[0]["data"] if prod.get(tname(op["inputs"][0])) != "Conv" else ins[0]["data"].transpose(0, 3, 1, 2), 2.0, 2.0)
            assert out["scale"] == ins[0]["scale"] and out["zp"] == ins[0]["zp"]
        elif t in ("Reshape", "Transpose") and ins[0] is not None:
            src = ins[0]["data"] if prod.get(tname(op["inputs"][0])) != "Conv" else ins[0]["data"].transpose(0, 3, 1, 2)
            if t == "Reshape":
                got = np.ascontiguousarray(src).reshape(out["data"].shape)
            else:
                got = src.transpose([int(v) for v in op["attrs"]["perm"].split(",")])
            assert out["scale"] == ins[0]["scale"] and out["zp"] == ins[0]["zp"]
        if got is None:
            continue
        bad = int((got.reshape(-1) != out["data"].reshape(-1)).sum())
        st_ = stats[t]
        st_[0] += 1; st_[1] += got.size; st_[2] += bad
        if bad:
            diff = np.abs(got.reshape(-1).astype(int) - out["data"].reshape(-1).astype(int))
            print(f"  {t} {op['name']}: {bad}/{got.size} codes differ, max |d| = {diff.max()}")
    return dict(stats)
