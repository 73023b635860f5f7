"""TEST INFRASTRUCTURE ONLY -- driver for the reference oracle (oracle/_ref/libonnxstream_ref.so = the unmodified
reference sources + oracle/xnn_shim.cpp, see oracle/Makefile).  Used by tests/, smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import time
from typing import Dict, Iterable, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(_HERE, "_ref", "libonnxstream_ref.so")


def available() -> bool:
    return os.path.exists(REF_LIB)


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (a 256-thread box with a
    16-CPU quota makes pthreadpool's 256 spinning workers crawl)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def run_model(model_dir: str, inputs: Dict[str, np.ndarray], fp16: bool = True, fuse_attention: bool = True,
              parts: int = 2, extra_outputs: Iterable[str] = (), threads: Optional[int] = None, ops_cache: bool = False,
              runs: int = 1, wp: str = "ram+nocache", return_times: bool = False):
    """One fresh reference Model per call (the reference's C API cannot re-push inputs once fp16 arithmetic is on)."""
    import sys
    sys.path.insert(0, os.path.dirname(_HERE))
    from onnxstream_amd.bindings import Model

    threads = threads or usable_cores()
    m = Model(REF_LIB, threads, wp)
    m.read_file(os.path.join(model_dir, "model.txt"))
    lib = m.lib
    lib.ref_set_attention_parts.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    lib.ref_set_attention_parts(m.handle, parts)
    for e in extra_outputs:
        m.add_extra_output(e)
    if ops_cache:
        m.set_use_ops_cache(True)
        m.set_use_next_op_cache(True)
    times = []
    outs = None
    for r in range(runs):
        m.set_use_fp16_arithmetic(False)      # inputs must enter as fp32 through model_add_tensor
        for k, v in inputs.items():
            m.add_tensor(k, np.ascontiguousarray(v, np.float32))
        m.set_use_fp16_arithmetic(fp16)
        m.set_fuse_ops_in_attention(fuse_attention)
        t0 = time.perf_counter()
        m.run()
        times.append(time.perf_counter() - t0)
        outs = {}
        for name in m.get_all_tensor_names():
            got = m.get_tensor(name)
            if got is not None:
                outs[name] = got[0]
        m.clear_tensors()
    m.close()
    return (outs, times) if return_times else outs
