"""ctypes binding over the C ABI of ``libosgpu.so`` (include/osgpu.h) -- one method per exported entry point.

Only plumbing lives here: numpy <-> device buffers and argument marshalling.  There is deliberately NO CPU fallback:
if the HIP library is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSGPU_LIB") or os.path.join(_HERE, "libosgpu.so")   # (OSGPU_LIB: A/B runs against another build, as the host library honours it)

U8, F16, F32, I64 = 1, 2, 3, 4
_NP2DT = {np.dtype(np.uint8): U8, np.dtype(np.float16): F16, np.dtype(np.float32): F32, np.dtype(np.int64): I64}
ACT_NONE, ACT_SILU, ACT_SIGMOID = 0, 1, 2
UN = dict(sigmoid=0, erf=1, sqrt=2, sin=3, cos=4, neg=5, pow=6, silu=7, gelu_erf=8)
BIN = dict(add=0, sub=1, mul=2, div=3)

EXPORTS = [
    "osg_device_count", "osg_init", "osg_destroy", "osg_last_error", "osg_device_name", "osg_stream", "osg_set_autotune", "osg_tune_misses",
    "osg_malloc", "osg_free", "osg_upload", "osg_upload_sync", "osg_host_register", "osg_host_unregister", "osg_upload_pinned", "osg_upload_pinned_async", "osg_copy_fence", "osg_download", "osg_copy", "osg_memset", "osg_sync",
    "osg_graph_begin", "osg_graph_end", "osg_graph_launch", "osg_graph_destroy", "osg_timer_start", "osg_timer_stop",
    "osg_conv2d_nhwc", "osg_conv2d_nhwc_rb", "osg_conv2d_nhwc_v", "osg_gemm", "osg_gemm_ln", "osg_gemm_rowstats", "osg_gemm_w8", "osg_conv2d_nhwc_w8", "osg_gemm_w8_v", "osg_conv2d_nhwc_w8_v", "osg_transpose_kn_to_nk", "osg_attention", "osg_attention_strided", "osg_sdpa", "osg_rms_norm", "osg_rope",
    "osg_instance_norm", "osg_group_norm_nhwc", "osg_layer_norm", "osg_reduce_mean_last", "osg_softmax_last",
    "osg_unary", "osg_binary", "osg_geglu", "osg_transpose", "osg_copy_2d", "osg_concat2", "osg_resize_nearest", "osg_gather_rows",
    "osg_maxpool_nhwc", "osg_convert", "osg_sampler_prepare", "osg_sampler_cfg_euler_a",
    "osg_range_push", "osg_range_pop", "osg_marker_record", "osg_copy_wait_marker", "osg_timer_mark", "osg_timer_between", "osg_set_stat_sinks", "osg_group_norm_stats_nhwc", "osg_qu8_conv2d_nhwc", "osg_qu8_conv2d_nhwc_t", "osg_qu8_conv_tap_sums", "osg_qu8_gemm", "osg_qu8_lut", "osg_qu8_binary", "osg_qu8_instance_norm", "osg_qu8_instance_norm_nhwc", "osg_qu8_affine_act", "osg_qu8_norm_affine_act_nhwc", "osg_qu8_softmax_last", "osg_kdbg_read",
    "osg_tblock_tail_supported", "osg_tblock_tail", "osg_tblock_kv_pack_elems", "osg_tblock_kv_pack_jobs", "osg_tblock_pack_weight", 
]


class OsgError(RuntimeError):
    pass


class TBlockTailArgs(ctypes.Structure):
    """osg_tblock_tail_args of include/osgpu.h"""
    _vp, _cf, _ci, _cl = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_long
    _fields_ = [("a1", _vp), ("x0", _vp), ("wo1", _vp), ("bo1", _vp), ("g2", _vp), ("be2", _vp), ("eps2", _cf), ("wq2", _vp), ("bq2", _vp),
                ("kp", _vp), ("vtp", _vp), ("scale", _cf), ("Tk", _ci), ("wo2", _vp), ("bo2", _vp), ("g3", _vp), ("be3", _vp), ("eps3", _cf),
                ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("wpo", _vp), ("bpo", _vp), ("xin", _vp), ("out", _vp), ("out2", _vp),
                ("ldo", _cl), ("ldo2", _cl), ("M", _ci), ("rows_per_img", _ci), ("C", _ci), ("heads", _ci), ("dbg", _vp * 8), ("rows_per_block", _ci)]


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise OsgError(f"{path} not found: build it with `python -m onnxstream_amd.build` (hipcc, gfx950)")
    lib = ctypes.CDLL(path)
    vp, ci, cl, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float
    lib.osg_init.argtypes = [ci, ctypes.POINTER(vp)]
    lib.osg_destroy.argtypes = [vp]
    lib.osg_destroy.restype = None
    lib.osg_last_error.argtypes = [vp]
    lib.osg_last_error.restype = ctypes.c_char_p
    lib.osg_device_name.argtypes = [vp]
    lib.osg_device_name.restype = ctypes.c_char_p
    lib.osg_stream.argtypes = [vp]
    lib.osg_stream.restype = vp
    lib.osg_malloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.osg_free.argtypes = [vp, vp]
    lib.osg_host_register.argtypes = [vp, vp, ctypes.c_size_t]
    lib.osg_host_unregister.argtypes = [vp, vp]
    for f in ("osg_upload", "osg_upload_sync", "osg_upload_pinned", "osg_upload_pinned_async", "osg_download", "osg_copy"):
        getattr(lib, f).argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.osg_memset.argtypes = [vp, vp, ci, ctypes.c_size_t]
    lib.osg_sync.argtypes = [vp]
    lib.osg_copy_fence.argtypes = [vp]
    lib.osg_graph_begin.argtypes = [vp]
    lib.osg_graph_end.argtypes = [vp, ctypes.POINTER(vp)]
    lib.osg_graph_launch.argtypes = [vp, vp]
    lib.osg_graph_destroy.argtypes = [vp]
    lib.osg_graph_destroy.restype = None
    lib.osg_timer_start.argtypes = [vp]
    lib.osg_timer_stop.argtypes = [vp, ctypes.POINTER(cf)]
    lib.osg_conv2d_nhwc.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp] + [ci] * 14
    lib.osg_conv2d_nhwc_rb.argtypes = [vp, ci, vp, vp, vp, ci, vp, cl, vp, vp] + [ci] * 14
    lib.osg_conv2d_nhwc_v.argtypes = [vp, ci, vp, vp, vp, ci, vp, cl, vp, vp, cl, vp, cl] + [ci] * 14
    lib.osg_gemm_w8.argtypes = [vp, vp, vp, cf, ci, vp, ci, vp, vp, ci, ci, ci, ci]
    lib.osg_conv2d_nhwc_w8.argtypes = [vp, vp, vp, cf, ci, vp, ci, vp, cl, vp, vp] + [ci] * 14
    lib.osg_gemm_w8_v.argtypes = [vp, vp, vp, cf, ci, vp, vp, vp, ci, vp, vp, ci, ci, ci, ci]
    lib.osg_conv2d_nhwc_w8_v.argtypes = [vp, vp, vp, cf, ci, vp, vp, vp, ci, vp, cl, vp, vp, cl, vp, cl] + [ci] * 14
    lib.osg_gemm.argtypes = [vp, ci, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, cl, cl, cl, ci]
    lib.osg_transpose_kn_to_nk.argtypes = [vp, ci, vp, vp, ci, ci]
    lib.osg_attention.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci]
    lib.osg_attention_strided.argtypes = [vp, ci, vp, cl, cl, cl, vp, cl, cl, cl, vp, cl, cl, cl, vp, cl, cl, cl, ci, ci, ci, ci,
                                          ci, cf]
    lib.osg_sdpa.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf]
    lib.osg_rms_norm.argtypes = [vp, ci, vp, vp, vp, cl, ci, cf]
    lib.osg_rope.argtypes = [vp, ci, vp, vp, vp, vp, cl, cl, ci]
    lib.osg_instance_norm.argtypes = [vp, ci, vp, vp, vp, vp, ci, cl, ci, cf]
    lib.osg_group_norm_nhwc.argtypes = [vp, ci, vp, vp, vp, vp, ci, cl, ci, ci, cf, ci]
    lib.osg_layer_norm.argtypes = [vp, ci, vp, vp, vp, vp, cl, ci, cf]
    lib.osg_reduce_mean_last.argtypes = [vp, ci, vp, vp, cl, cl]
    lib.osg_softmax_last.argtypes = [vp, ci, vp, vp, cl, cl]
    lib.osg_unary.argtypes = [vp, ci, ci, vp, vp, cl, cf]
    lib.osg_binary.argtypes = [vp, ci, ci, vp, ctypes.POINTER(cl), vp, ctypes.POINTER(cl), vp, ci]
    lib.osg_geglu.argtypes = [vp, ci, vp, vp, cl, cl]
    lib.osg_transpose.argtypes = [vp, ci, vp, vp, ci, ctypes.POINTER(cl), ctypes.POINTER(ci)]
    lib.osg_copy_2d.argtypes = [vp, ci, vp, cl, cl, vp, cl, cl, cl, cl]
    lib.osg_resize_nearest.argtypes = [vp, ci, vp, vp] + [ci] * 7
    lib.osg_gather_rows.argtypes = [vp, ci, vp, vp, vp, cl, cl, cl]
    lib.osg_maxpool_nhwc.argtypes = [vp, ci, vp, vp] + [ci] * 12
    lib.osg_convert.argtypes = [vp, ci, ci, vp, vp, cl, cf, ci]
    lib.osg_concat2.argtypes = [vp, ci, vp, cl, vp, cl, vp, cl]
    lib.osg_gemm_ln.argtypes = [vp, vp, vp, vp, vp, cf, vp, vp, vp, ci, ci, ci, ci]
    lib.osg_gemm_rowstats.argtypes = [vp, vp, vp, vp, ci, vp, vp, ci, ci, ci, ci, vp]
    lib.osg_sampler_prepare.argtypes = [vp, vp, vp, vp, ci, cl, cf, cf, cl]
    lib.osg_sampler_cfg_euler_a.argtypes = [vp, vp, vp, vp, ci, cl, cf, cf, cf, cf, cf, cf]
    lib.osg_set_stat_sinks.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, ci, ci]
    lib.osg_group_norm_stats_nhwc.argtypes = [vp, vp, vp, vp, vp, ci, ctypes.c_long, ci, ci, cf, ci, vp]
    lib.osg_qu8_conv2d_nhwc.argtypes = [vp, vp, cf, ci, vp, cf, ci, vp, cf, ci, vp] + [ci] * 13
    lib.osg_qu8_conv2d_nhwc_t.argtypes = [vp, vp, cf, ci, vp, cf, ci, vp, cf, ci, vp] + [ci] * 13 + [vp]
    lib.osg_qu8_conv_tap_sums.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    lib.osg_qu8_gemm.argtypes = [vp, vp, cl, cf, ci, vp, cf, ci, vp, cf, ci, vp, ci, ci, ci, ci, cl, cl, cl]
    lib.osg_qu8_lut.argtypes = [vp, vp, vp, cl, vp]
    lib.osg_qu8_binary.argtypes = [vp, ci, vp, ctypes.POINTER(cl), cf, ci, vp, ctypes.POINTER(cl), cf, ci, vp, cf, ci, ci]
    lib.osg_qu8_instance_norm.argtypes = [vp, vp, vp, ci, cl, ci, vp, vp, cf, cf, ci, cf, ci]
    lib.osg_qu8_affine_act.argtypes = [vp, vp, cf, ci, vp, cf, ci, cf, ci, vp, cf, ci, cf, ci, vp, cf, ci, cf, ci, vp, cl, ci, cl]
    lib.osg_qu8_instance_norm_nhwc.argtypes = [vp, vp, vp, cl, ci, ci, ci, vp, vp, cf, cf, ci, cf, ci]
    lib.osg_qu8_norm_affine_act_nhwc.argtypes = [vp, vp, cl, ci, ci, ci, vp, vp, cf, cf, ci, cf, ci, vp, cf, ci, cf, ci, vp, cf, ci, cf, ci, vp, cf, ci, cf, ci, vp]
    lib.osg_qu8_softmax_last.argtypes = [vp, vp, vp, cl, cl, vp]
    lib.osg_tblock_tail_supported.argtypes = [ci] * 5
    lib.osg_tblock_tail.argtypes = [vp, ctypes.POINTER(TBlockTailArgs)]
    lib.osg_tblock_kv_pack_elems.argtypes = [ci, ci, ci]
    lib.osg_tblock_kv_pack_elems.restype = ctypes.c_size_t
    lib.osg_tblock_kv_pack_jobs.argtypes = [vp, vp, cl, ci, ci, ci, ci, vp, vp]
    lib.osg_tblock_pack_weight.argtypes = [vp, vp, ci, ci, vp]
    return lib


class DevBuf:
    """A device allocation with numpy-side shape/dtype metadata."""

    def __init__(self, gpu: "Gpu", shape, dtype):
        self.gpu = gpu
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        gpu._ck(gpu.lib.osg_malloc(gpu.ctx, max(self.nbytes, 16), ctypes.byref(p)))
        self.ptr = p.value

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            self.gpu._ck(self.gpu.lib.osg_download(self.gpu.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.gpu.lib.osg_free(self.gpu.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Gpu:
    def __init__(self, device: int = 0, lib_path: str = LIB_PATH):
        self.lib = load_library(lib_path)
        if self.lib.osg_device_count() <= 0:
            raise OsgError("no HIP device visible: the osgpu backend has no CPU fallback")
        c = ctypes.c_void_p()
        rc = self.lib.osg_init(device, ctypes.byref(c))
        if rc:
            raise OsgError(f"osg_init failed with code {rc}")
        self.ctx = c

    def close(self):
        if self.ctx:
            self.lib.osg_destroy(self.ctx)
            self.ctx = None

    def _ck(self, rc):
        if rc:
            raise OsgError(self.lib.osg_last_error(self.ctx).decode())

    @property
    def name(self):
        return self.lib.osg_device_name(self.ctx).decode()

    # ---- memory ----
    def empty(self, shape, dtype=np.float16) -> DevBuf:
        return DevBuf(self, shape, dtype)

    def to_dev(self, arr: np.ndarray, staged: bool = False) -> DevBuf:
        arr = np.ascontiguousarray(arr)
        b = DevBuf(self, arr.shape, arr.dtype)
        if arr.nbytes:
            fn = self.lib.osg_upload if staged else self.lib.osg_upload_sync
            self._ck(fn(self.ctx, b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def sync(self):
        self._ck(self.lib.osg_sync(self.ctx))

    def timer_start(self):
        self._ck(self.lib.osg_timer_start(self.ctx))

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        self._ck(self.lib.osg_timer_stop(self.ctx, ctypes.byref(ms)))
        return ms.value

    # ---- compute ----
    @staticmethod
    def _p(b: Optional[DevBuf]):
        return b.ptr if b is not None else None

    def conv2d_nhwc(self, x: DevBuf, w: DevBuf, bias: Optional[DevBuf], stride=1, pads=(1, 1, 1, 1), residual=None, act=ACT_NONE,
                    image_bias: Optional[DevBuf] = None):
        n, h, wd, cin = x.shape
        cout, kh, kw, cin2 = w.shape
        assert cin == cin2
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        pt, pl, pb, pr = pads
        ho, wo = (h + pt + pb - kh) // sh + 1, (wd + pl + pr - kw) // sw + 1
        y = self.empty((n, ho, wo, cout), x.dtype)
        bdt = _NP2DT[bias.dtype] if bias is not None else F16
        if image_bias is not None:   # [n, cout] f16, added per image (fused resnet time-embedding add)
            self._ck(self.lib.osg_conv2d_nhwc_rb(self.ctx, _NP2DT[x.dtype], x.ptr, w.ptr, self._p(bias), bdt, image_bias.ptr, cout,
                                                 self._p(residual), y.ptr, n, h, wd, cin, cout, kh, kw, sh, sw, pt, pl, pb, pr, act))
            return y
        self._ck(self.lib.osg_conv2d_nhwc(self.ctx, _NP2DT[x.dtype], x.ptr, w.ptr, self._p(bias), bdt, self._p(residual), y.ptr, n, h,
                                          wd, cin, cout, kh, kw, sh, sw, pt, pl, pb, pr, act))
        return y

    def conv2d_nhwc_view(self, x: DevBuf, w: DevBuf, bias: Optional[DevBuf], wide: DevBuf, col: int, dense: Optional[DevBuf] = None, stride=1,
                         pads=(1, 1, 1, 1), residual=None, act=ACT_NONE, image_bias: Optional[DevBuf] = None, swap=False):
        """osg_conv2d_nhwc_v: the result goes into columns [col, col + Cout) of the NHWC buffer `wide` (its last axis is the row pitch) and, when
        `dense` is given, a second time into that dense [n, ho, wo, Cout] tensor (swap: dense is the primary destination, the slice the second)."""
        n, h, wd, cin = x.shape
        cout, kh, kw, _ = w.shape
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        pt, pl, pb, pr = pads
        bdt = _NP2DT[bias.dtype] if bias is not None else F16
        es = x.dtype.itemsize
        v_ptr, v_ld = wide.ptr + col * es, wide.shape[-1]
        if dense is None:
            y, y_ld, y2, y2_ld = v_ptr, v_ld, None, 0
        elif swap:
            y, y_ld, y2, y2_ld = dense.ptr, 0, v_ptr, v_ld
        else:
            y, y_ld, y2, y2_ld = v_ptr, v_ld, dense.ptr, cout
        self._ck(self.lib.osg_conv2d_nhwc_v(self.ctx, _NP2DT[x.dtype], x.ptr, w.ptr, self._p(bias), bdt, self._p(image_bias), cout if image_bias is not None else 0,
                                            self._p(residual), y, y_ld, y2, y2_ld, n, h, wd, cin, cout, kh, kw, sh, sw, pt, pl, pb, pr, act))

    def gemm(self, a: DevBuf, b: DevBuf, bias=None, residual=None, b_is_nk=False, act=ACT_NONE):
        """a:[(batch,)M,K]; b:[(batch,)K,N] (or [(batch,)N,K] when b_is_nk)."""
        batch = a.shape[0] if len(a.shape) == 3 else 1
        m, k = a.shape[-2:]
        n = b.shape[-2] if b_is_nk else b.shape[-1]
        sb = 0 if len(b.shape) == 2 else k * n
        c = self.empty(a.shape[:-1] + (n,), a.dtype)
        bdt = _NP2DT[bias.dtype] if bias is not None else F16
        self._ck(self.lib.osg_gemm(self.ctx, _NP2DT[a.dtype], a.ptr, b.ptr, int(b_is_nk), self._p(bias), bdt, self._p(residual), c.ptr, m,
                                   n, k, batch, m * k if batch > 1 else 0, sb, m * n if batch > 1 else 0, act))
        return c

    def gemm_rowstats(self, a: DevBuf, b_nk: DevBuf, bias=None, residual=None, act=ACT_NONE):
        """osg_gemm_rowstats: returns (C, rowstats[M, N/32, 2])."""
        m, k = a.shape
        n = b_nk.shape[0]
        c = self.empty((m, n), a.dtype)
        rs = self.empty((m, n // 32, 2), np.dtype(np.float32))
        bdt = _NP2DT[bias.dtype] if bias is not None else F16
        self._ck(self.lib.osg_gemm_rowstats(self.ctx, a.ptr, b_nk.ptr, self._p(bias), bdt, self._p(residual), c.ptr, m, n, k, act, rs.ptr))
        return c, rs

    def gemm_ln(self, x: DevBuf, w_nk: np.ndarray, gamma: np.ndarray, beta: np.ndarray, bias=None, eps: float = 1e-5, residual=None, act=ACT_NONE,
                rowstats=None):
        """LayerNorm(x; gamma, beta, eps) . w_nk^T + bias through osg_gemm_ln: folds gamma into the [N,K] weight and builds c1 / c2 on the
        host exactly as the planner does (plan.cpp ln_fold_weight).  x:[M,K] device f16; w_nk, gamma, beta, bias: host f16 arrays."""
        m, k = x.shape
        n = w_nk.shape[0]
        wf = (gamma.astype(np.float32)[None, :] * w_nk.astype(np.float32)).astype(np.float16)
        c1 = wf.astype(np.float64).sum(axis=1).astype(np.float32)
        c2 = (w_nk.astype(np.float64) @ beta.astype(np.float64) + (bias.astype(np.float64) if bias is not None else 0.0)).astype(np.float32)
        dw, d1, d2 = self.to_dev(wf), self.to_dev(c1), self.to_dev(c2)
        y = self.empty((m, n // 2 if act == 3 else n), x.dtype)
        self._ck(self.lib.osg_gemm_ln(self.ctx, x.ptr, dw.ptr, d1.ptr, d2.ptr, eps, self._p(rowstats), self._p(residual), y.ptr, m, n, k, act))
        return y

    def transpose_kn_to_nk(self, w: DevBuf):
        k, n = w.shape
        o = self.empty((n, k), w.dtype)
        self._ck(self.lib.osg_transpose_kn_to_nk(self.ctx, _NP2DT[w.dtype], w.ptr, o.ptr, k, n))
        return o

    def attention(self, q: DevBuf, k: DevBuf, v: DevBuf, scale: float, k_is_dt: bool):
        heads, tq, d = q.shape
        tkv = v.shape[1]
        o = self.empty(q.shape, q.dtype)
        self._ck(self.lib.osg_attention(self.ctx, _NP2DT[q.dtype], q.ptr, k.ptr, v.ptr, o.ptr, heads, tq, tkv, d, scale, int(k_is_dt)))
        return o

    def attention_tokens(self, q: DevBuf, k: DevBuf, v: DevBuf, heads: int, scale: float):
        """q:[B,Tq,heads*D], k,v:[B,Tkv,heads*D] straight out of the projections -> o:[B,Tq,heads*D]."""
        bsz, tq, c = q.shape
        tkv = k.shape[1]
        d = c // heads
        o = self.empty(q.shape, q.dtype)
        self._ck(self.lib.osg_attention_strided(self.ctx, F16, q.ptr, c, d, tq * c, k.ptr, c, d, tkv * c, v.ptr, c, d, tkv * c, o.ptr, c, d,
                                                tq * c, bsz, heads, tq, tkv, d, scale))
        return o

    def tblock_kv_pack(self, k: DevBuf, v: DevBuf, heads: int):
        """k, v: [imgs, Tk, heads*D] -> (kp [imgs, heads, 80, DP], vtp [imgs, heads, DP, 80]) for tblock_tail.  Goes through the multi-job entry point the
        planner uses: K and V become two column ranges of one [imgs*Tk, 2C] matrix."""
        imgs, tk, c = k.shape
        d = c // heads
        dp = (d + 15) // 16 * 16
        n = self.lib.osg_tblock_kv_pack_elems(imgs, heads, d)
        assert n == imgs * heads * 80 * dp
        kv = self.to_dev(np.concatenate([k.numpy().reshape(imgs * tk, c), v.numpy().reshape(imgs * tk, c)], axis=1))
        jobs = self.to_dev(np.array([0, c, d, 0], np.int32))
        dst = self.empty((2 * n,), k.dtype)
        self._ck(self.lib.osg_tblock_kv_pack_jobs(self.ctx, kv.ptr, 2 * c, imgs, tk, heads, 1, jobs.ptr, dst.ptr))
        kp, vtp = self.empty((imgs, heads, 80, dp), k.dtype), self.empty((imgs, heads, dp, 80), k.dtype)
        self._ck(self.lib.osg_copy(self.ctx, kp.ptr, dst.ptr, n * 2))
        self._ck(self.lib.osg_copy(self.ctx, vtp.ptr, dst.ptr + n * 2, n * 2))
        return kp, vtp

    TBLOCK_WEIGHTS = ("wo1", "wq2", "wo2", "w1", "w2", "wpo")

    def tblock_pack_weight(self, w_nk: DevBuf) -> DevBuf:
        """[N, K] (k contiguous) -> the kn8 layout osg_tblock_tail streams, returned as a [K/8, N, 8] buffer"""
        n, k = w_nk.shape
        out = self.empty((k // 8, n, 8), w_nk.dtype)
        self._ck(self.lib.osg_tblock_pack_weight(self.ctx, w_nk.ptr, n, k, out.ptr))
        return out

    def tblock_weights(self, w: dict) -> dict:
        """host dict of a block's operands (weights [N, K]) -> device dict as tblock_tail wants it (weights packed)"""
        return {n: (self.tblock_pack_weight(self.to_dev(t)) if n in self.TBLOCK_WEIGHTS else self.to_dev(t)) for n, t in w.items()}

    def tblock_tail(self, a1: DevBuf, x0: DevBuf, w: dict, kp: DevBuf, vtp: DevBuf, tk: int, heads: int, scale: float, rows_per_img: int, eps: float = 1e-5,
                    xin: Optional[DevBuf] = None, out2: Optional[DevBuf] = None, out2_col: int = 0, debug: bool = False, out: Optional[DevBuf] = None, stamps: Optional[DevBuf] = None, rows_per_block: int = 0):
        """osg_tblock_tail.  w: dict of DevBuf -- wo1 bo1 g2 be2 wq2 wo2 bo2 g3 be3 w1 b1 w2 b2 [wpo bpo]; weights in the kn8 layout (tblock_weights).  Returns (out, [dumps])."""
        m, c = a1.shape
        a = TBlockTailArgs()
        a.a1, a.x0 = a1.ptr, x0.ptr
        for k_ in ("wo1", "bo1", "g2", "be2", "wq2", "wo2", "bo2", "g3", "be3", "w1", "b1", "w2", "b2", "wpo", "bpo"):
            setattr(a, k_, w[k_].ptr if w.get(k_) is not None else None)
        a.bq2 = None
        a.kp, a.vtp, a.scale, a.Tk = kp.ptr, vtp.ptr, scale, tk
        a.eps2 = a.eps3 = eps
        a.xin = xin.ptr if xin is not None else None
        if out is None:
            out = self.empty((m, c), a1.dtype)
        a.out, a.ldo = out.ptr, c
        if out2 is not None:
            a.out2, a.ldo2 = out2.ptr + out2_col * 2, out2.shape[-1]
        a.M, a.rows_per_img, a.C, a.heads = m, rows_per_img, c, heads
        a.rows_per_block = rows_per_block
        dumps = []
        if debug:
            dumps = [self.empty((m, c), a1.dtype) for _ in range(7)]
            for i, dbuf in enumerate(dumps):
                a.dbg[i] = dbuf.ptr
        if stamps is not None:
            a.dbg[7] = stamps.ptr
        self._ck(self.lib.osg_tblock_tail(self.ctx, ctypes.byref(a)))
        return out, dumps

    def rms_norm(self, x: DevBuf, w: DevBuf, eps: float):
        rows, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_rms_norm(self.ctx, _NP2DT[x.dtype], x.ptr, w.ptr, y.ptr, rows, c, eps))
        return y

    def rope(self, x: DevBuf, cos: DevBuf, sin: DevBuf):
        """x:[..., T, d], cos / sin:[T, d]"""
        t, d = x.shape[-2], x.shape[-1]
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_rope(self.ctx, _NP2DT[x.dtype], x.ptr, cos.ptr, sin.ptr, y.ptr, int(np.prod(x.shape[:-2])), t, d))
        return y

    def sdpa(self, q: DevBuf, k: DevBuf, v: DevBuf, mask: Optional[DevBuf], scale: float):
        """q:[B,Hq,Tq,D], k,v:[B,Hkv,Tkv,D], mask:[Tq,Tkv] additive or None -> o:[B,Hq,Tq,D] (the reference's ScaledDotProductAttention op)."""
        bsz, hq, tq, d = q.shape
        hkv, tkv = k.shape[1], k.shape[2]
        o = self.empty(q.shape, q.dtype)
        self._ck(self.lib.osg_sdpa(self.ctx, _NP2DT[q.dtype], q.ptr, k.ptr, v.ptr, self._p(mask), o.ptr, bsz, hq, hkv, tq, tkv, d, scale))
        return o

    def instance_norm(self, x: DevBuf, scale: Optional[DevBuf], bias: Optional[DevBuf], eps: float):
        rows, L = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, x.dtype)
        ns = scale.size if scale is not None else 1
        self._ck(self.lib.osg_instance_norm(self.ctx, _NP2DT[x.dtype], x.ptr, self._p(scale), self._p(bias), y.ptr, rows, L, ns, eps))
        return y

    def set_stat_sinks(self, rows_per_image: int, t0: Optional[DevBuf] = None, groups0=0, cpg0=0, off0=0, t1: Optional[DevBuf] = None, groups1=0, cpg1=0, off1=0):
        """osg_set_stat_sinks: arm the NEXT conv2d_nhwc / conv2d_nhwc_view launch (tables: int64 [N][groups][2], zeroed by the caller)."""
        self._ck(self.lib.osg_set_stat_sinks(self.ctx, self._p(t0), groups0, cpg0, off0, self._p(t1), groups1, cpg1, off1, rows_per_image))

    def group_norm_stats_nhwc(self, x: DevBuf, gamma: DevBuf, beta: DevBuf, groups: int, eps: float, table: DevBuf, act=ACT_NONE):
        n, h, w, c = x.shape
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_group_norm_stats_nhwc(self.ctx, x.ptr, gamma.ptr, beta.ptr, y.ptr, n, h * w, c, groups, eps, act, table.ptr))
        return y

    def group_norm_nhwc(self, x: DevBuf, gamma: DevBuf, beta: DevBuf, groups: int, eps: float, act=ACT_NONE):
        n, h, w, c = x.shape
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_group_norm_nhwc(self.ctx, _NP2DT[x.dtype], x.ptr, gamma.ptr, beta.ptr, y.ptr, n, h * w, c, groups, eps, act))
        return y

    def layer_norm(self, x: DevBuf, gamma: DevBuf, beta: DevBuf, eps: float):
        rows, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_layer_norm(self.ctx, _NP2DT[x.dtype], x.ptr, gamma.ptr, beta.ptr, y.ptr, rows, c, eps))
        return y

    def reduce_mean_last(self, x: DevBuf):
        rows, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape[:-1] + (1,), x.dtype)
        self._ck(self.lib.osg_reduce_mean_last(self.ctx, _NP2DT[x.dtype], x.ptr, y.ptr, rows, c))
        return y

    def softmax_last(self, x: DevBuf):
        rows, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_softmax_last(self.ctx, _NP2DT[x.dtype], x.ptr, y.ptr, rows, c))
        return y

    def unary(self, kind: str, x: DevBuf, param: float = 0.0):
        y = self.empty(x.shape, x.dtype)
        self._ck(self.lib.osg_unary(self.ctx, _NP2DT[x.dtype], UN[kind], x.ptr, y.ptr, x.size, param))
        return y

    def binary(self, kind: str, a: DevBuf, b: DevBuf):
        rank = max(len(a.shape), len(b.shape))
        ash = (1,) * (rank - len(a.shape)) + a.shape
        bsh = (1,) * (rank - len(b.shape)) + b.shape
        osh = tuple(np.broadcast_shapes(ash, bsh))
        y = self.empty(osh, a.dtype)
        A = (ctypes.c_long * rank)(*ash)
        B = (ctypes.c_long * rank)(*bsh)
        self._ck(self.lib.osg_binary(self.ctx, _NP2DT[a.dtype], BIN[kind], a.ptr, A, b.ptr, B, y.ptr, rank))
        return y

    def geglu(self, x: DevBuf):
        rows, c2 = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape[:-1] + (c2 // 2,), x.dtype)
        self._ck(self.lib.osg_geglu(self.ctx, _NP2DT[x.dtype], x.ptr, y.ptr, rows, c2 // 2))
        return y

    def transpose(self, x: DevBuf, perm: Sequence[int]):
        rank = len(x.shape)
        y = self.empty(tuple(x.shape[p] for p in perm), x.dtype)
        S = (ctypes.c_long * rank)(*x.shape)
        P = (ctypes.c_int * rank)(*perm)
        self._ck(self.lib.osg_transpose(self.ctx, x.dtype.itemsize, x.ptr, y.ptr, rank, S, P))
        return y

    def copy_2d(self, src: DevBuf, src_pitch, src_off, dst: DevBuf, dst_pitch, dst_off, outer, inner):
        self._ck(self.lib.osg_copy_2d(self.ctx, src.dtype.itemsize, src.ptr, src_pitch, src_off, dst.ptr, dst_pitch, dst_off, outer, inner))

    def concat2(self, a: DevBuf, b: DevBuf):
        """Concat of two dense tensors along the last axis in one launch."""
        outer = int(np.prod(a.shape[:-1]))
        y = self.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), a.dtype)
        self._ck(self.lib.osg_concat2(self.ctx, a.dtype.itemsize, a.ptr, a.shape[-1], b.ptr, b.shape[-1], y.ptr, outer))
        return y

    def resize_nearest(self, x: DevBuf, ho: int, wo: int, nhwc: bool):
        if nhwc:
            n, h, w, c = x.shape
            y = self.empty((n, ho, wo, c), x.dtype)
        else:
            n, c, h, w = x.shape
            y = self.empty((n, c, ho, wo), x.dtype)
        self._ck(self.lib.osg_resize_nearest(self.ctx, x.dtype.itemsize, x.ptr, y.ptr, n, c, h, w, ho, wo, int(nhwc)))
        return y

    def gather_rows(self, x: DevBuf, idx: DevBuf):
        n_rows = x.shape[0]
        row = int(np.prod(x.shape[1:]))
        y = self.empty((idx.size,) + x.shape[1:], x.dtype)
        self._ck(self.lib.osg_gather_rows(self.ctx, x.dtype.itemsize, x.ptr, idx.ptr, y.ptr, idx.size, row, n_rows))
        return y

    def maxpool_nhwc(self, x: DevBuf, k, stride, pads):
        n, h, w, c = x.shape
        pt, pl, pb, pr = pads
        ho, wo = (h + pt + pb - k[0]) // stride[0] + 1, (w + pl + pr - k[1]) // stride[1] + 1
        y = self.empty((n, ho, wo, c), x.dtype)
        self._ck(self.lib.osg_maxpool_nhwc(self.ctx, _NP2DT[x.dtype], x.ptr, y.ptr, n, h, w, c, k[0], k[1], stride[0], stride[1], pt, pl,
                                           pb, pr))
        return y

    def convert(self, x: DevBuf, dtype, scale: float = 1.0, zero_point: int = 0):
        y = self.empty(x.shape, dtype)
        self._ck(self.lib.osg_convert(self.ctx, _NP2DT[x.dtype], _NP2DT[np.dtype(dtype)], x.ptr, y.ptr, x.size, scale, zero_point))
        return y

    # ---- uint8 arithmetic (a uint8 tensor = codes DevBuf + (scale, zero_point)) ----
    def qu8_conv_tap_sums(self, w: DevBuf) -> DevBuf:
        cout, kh, kw, cin = w.shape
        t = self.empty((cout, kh * kw), np.int32)
        self._ck(self.lib.osg_qu8_conv_tap_sums(self.ctx, w.ptr, cout, kh, kw, cin, t.ptr))
        return t

    def qu8_conv2d_nhwc(self, x: DevBuf, xq, w: DevBuf, wq, bias: Optional[DevBuf], oq, stride=1, pads=(1, 1, 1, 1), tap_sums: Optional[DevBuf] = None):
        n, h, wd, cin = x.shape
        cout, kh, kw, _ = w.shape
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        pt, pl, pb, pr = pads
        ho, wo = (h + pt + pb - kh) // sh + 1, (wd + pl + pr - kw) // sw + 1
        y = self.empty((n, ho, wo, cout), np.uint8)
        if tap_sums is not None:
            self._ck(self.lib.osg_qu8_conv2d_nhwc_t(self.ctx, x.ptr, float(xq[0]), int(xq[1]), w.ptr, float(wq[0]), int(wq[1]), self._p(bias), float(oq[0]), int(oq[1]),
                                                    y.ptr, n, h, wd, cin, cout, kh, kw, sh, sw, pt, pl, pb, pr, tap_sums.ptr))
            return y
        self._ck(self.lib.osg_qu8_conv2d_nhwc(self.ctx, x.ptr, float(xq[0]), int(xq[1]), w.ptr, float(wq[0]), int(wq[1]), self._p(bias), float(oq[0]), int(oq[1]),
                                              y.ptr, n, h, wd, cin, cout, kh, kw, sh, sw, pt, pl, pb, pr))
        return y

    def qu8_gemm(self, a: DevBuf, aq, b_nk: DevBuf, bq, bias: Optional[DevBuf], oq):
        """a:[(batch,)M,K], b_nk:[(batch,)N,K] codes."""
        batch = a.shape[0] if len(a.shape) == 3 else 1
        m, k = a.shape[-2:]
        n = b_nk.shape[-2]
        c = self.empty(a.shape[:-1] + (n,), np.uint8)
        self._ck(self.lib.osg_qu8_gemm(self.ctx, a.ptr, k, float(aq[0]), int(aq[1]), b_nk.ptr, float(bq[0]), int(bq[1]), self._p(bias), float(oq[0]), int(oq[1]),
                                       c.ptr, m, n, k, batch, m * k if batch > 1 else 0, n * k if len(b_nk.shape) == 3 else 0, m * n if batch > 1 else 0))
        return c

    def qu8_lut(self, x: DevBuf, lut: np.ndarray):
        y = self.empty(x.shape, np.uint8)
        t = self.to_dev(np.ascontiguousarray(lut, np.uint8))
        self._ck(self.lib.osg_qu8_lut(self.ctx, x.ptr, y.ptr, x.size, t.ptr))
        self.sync()
        return y

    def qu8_binary(self, kind: str, a: DevBuf, aq, b: DevBuf, bq, oq):
        rank = max(len(a.shape), len(b.shape))
        ash = (1,) * (rank - len(a.shape)) + a.shape
        bsh = (1,) * (rank - len(b.shape)) + b.shape
        y = self.empty(tuple(np.broadcast_shapes(ash, bsh)), np.uint8)
        A = (ctypes.c_long * rank)(*ash)
        B = (ctypes.c_long * rank)(*bsh)
        self._ck(self.lib.osg_qu8_binary(self.ctx, BIN[kind], a.ptr, A, float(aq[0]), int(aq[1]), b.ptr, B, float(bq[0]), int(bq[1]), y.ptr, float(oq[0]), int(oq[1]), rank))
        return y

    def qu8_instance_norm(self, x: DevBuf, xq, scale: DevBuf, bias: DevBuf, eps: float, oq):
        rows, L = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, np.uint8)
        self._ck(self.lib.osg_qu8_instance_norm(self.ctx, x.ptr, y.ptr, rows, L, scale.size, scale.ptr, bias.ptr, eps, float(xq[0]), int(xq[1]), float(oq[0]), int(oq[1])))
        return y

    def qu8_affine_act(self, x: DevBuf, xq, g: DevBuf, gq, mq, b: DevBuf, bq, aq, sig_lut: Optional[DevBuf], sq, oq, channels: int, inner: int = 1):
        """Mul(x, g[C]) -> Add(., b[C]) [-> Sigmoid -> Mul] in one pass; *q = (scale, zero point) of x, g, Mul out, b, Add out, Sigmoid out, final Mul out"""
        y = self.empty(x.shape, np.uint8)
        self._ck(self.lib.osg_qu8_affine_act(self.ctx, x.ptr, float(xq[0]), int(xq[1]), g.ptr, float(gq[0]), int(gq[1]), float(mq[0]), int(mq[1]), b.ptr, float(bq[0]),
                                             int(bq[1]), float(aq[0]), int(aq[1]), self._p(sig_lut), float(sq[0]), int(sq[1]), float(oq[0]), int(oq[1]), y.ptr, x.size,
                                             channels, inner))
        return y

    def qu8_instance_norm_nhwc(self, x: DevBuf, groups: int, xq, scale: DevBuf, bias: DevBuf, eps: float, oq):
        """x: [HW, C] codes (NHWC); rows of the normalisation = `groups` blocks of C/groups channels over every pixel"""
        hw, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, np.uint8)
        self._ck(self.lib.osg_qu8_instance_norm_nhwc(self.ctx, x.ptr, y.ptr, hw, c, groups, scale.size, scale.ptr, bias.ptr, eps, float(xq[0]), int(xq[1]), float(oq[0]), int(oq[1])))
        return y

    def qu8_norm_affine_act_nhwc(self, x: DevBuf, groups: int, xq, scale: DevBuf, bias: DevBuf, eps: float, nq, g: DevBuf, gq, mq, b: DevBuf, bq, aq, sig_lut: Optional[DevBuf], sq, oq):
        """qu8_instance_norm_nhwc followed by qu8_affine_act, the normalisation's per-group table applied inside the affine pass"""
        hw, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, np.uint8)
        self._ck(self.lib.osg_qu8_norm_affine_act_nhwc(self.ctx, x.ptr, hw, c, groups, scale.size, scale.ptr, bias.ptr, eps, float(xq[0]), int(xq[1]), float(nq[0]), int(nq[1]),
                                                       g.ptr, float(gq[0]), int(gq[1]), float(mq[0]), int(mq[1]), b.ptr, float(bq[0]), int(bq[1]), float(aq[0]), int(aq[1]),
                                                       self._p(sig_lut), float(sq[0]), int(sq[1]), float(oq[0]), int(oq[1]), y.ptr))
        return y

    def qu8_softmax_last(self, x: DevBuf, lut_u32: np.ndarray):
        rows, c = int(np.prod(x.shape[:-1])), x.shape[-1]
        y = self.empty(x.shape, np.uint8)
        t = self.to_dev(np.ascontiguousarray(lut_u32, np.uint32))
        self._ck(self.lib.osg_qu8_softmax_last(self.ctx, x.ptr, y.ptr, rows, c, t.ptr))
        self.sync()
        return y
