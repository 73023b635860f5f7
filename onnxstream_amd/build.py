"""In-tree native builds (no JIT cache: the built .so files travel to the GPU box with the snapshot).

  libosgpu.so          hipcc --offload-arch=gfx950   onnxstream_amd/csrc/osg_*.hip      (C ABI: include/osgpu.h)
  libonnxstream_amd.so g++                           onnxstream_amd/csrc/host/*.cpp      (Model host + model_* C API)
  oracle/_ref/...      make -C oracle ref            (only where /root/reference exists; test infrastructure)
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
INC = os.path.join(REPO, "include")
LIB_GPU = os.path.join(ROOT, "libosgpu.so")
LIB_HOST = os.environ.get("OSA_LIB_HOST") or os.path.join(ROOT, "libonnxstream_amd.so")   # (OSA_LIB_HOST: A/B runs against another build of the host library, tools/r3_ab_r2.sh)
ORACLE_REF = os.path.join(REPO, "oracle", "_ref", "libonnxstream_ref.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_gpu(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "osg_*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(INC, "osgpu.h")]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form", "-I" + INC, "-I" + CSRC, "-c", s, "-o", o]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("build failed: " + " ".join(cmd))
        if verbose and out:
            print(out)
    if force or procs or _newer(LIB_GPU, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_GPU] + objs)
    return LIB_GPU


def build_host(force=False):
    hdir = os.path.join(CSRC, "host")
    srcs = sorted(glob.glob(os.path.join(hdir, "*.cpp")))
    if not srcs:
        return None
    hdrs = sorted(glob.glob(os.path.join(hdir, "*.h"))) + sorted(glob.glob(os.path.join(hdir, "*.inc"))) + [os.path.join(INC, "osgpu.h")]
    if force or _newer(LIB_HOST, srcs + hdrs):
        _run(["g++", "-std=c++20", "-O2", "-fPIC", "-shared", "-I" + INC, "-I" + hdir, "-o", LIB_HOST] + srcs +
             ["-ldl", "-lpthread"])
    return LIB_HOST


def build_oracle_ref():
    """Reference oracle; only buildable where the reference sources are mounted."""
    if os.path.isdir("/root/reference/src"):
        _run(["make", "-C", os.path.join(REPO, "oracle"), "ref"])
    return ORACLE_REF if os.path.exists(ORACLE_REF) else None


def build_all(force=False):
    build_gpu(force)
    build_host(force)
    build_oracle_ref()


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built:", LIB_GPU, LIB_HOST if os.path.exists(LIB_HOST) else "(no host yet)")
