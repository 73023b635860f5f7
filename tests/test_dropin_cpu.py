"""CPU: the drop-in contract of the host header.  The reference's own translation units that sit ABOVE `class Model` --
src/exports.cpp (the C API the Python / C# / JS bindings load), src/sd.cpp (the txt2img application: every option field, WeightsProvider
template, CudaOptions and Tensor accessor it touches) and src/llm.cpp -- must compile UNCHANGED against
onnxstream_amd/csrc/host/onnxstream.h.  The sources are compiled where they lie (a COPY of each .cpp is placed next to our header in a
temp dir so that its `#include "onnxstream.h"` binds to ours; nothing of the reference enters the repository).  Needs /root/reference:
skipped on the GPU box."""
import os
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
HOST = os.path.join(REPO, "onnxstream_amd", "csrc", "host")


def _torch_include():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "include")     # cpuinfo.h (sd.cpp includes it for its core count)


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="needs the reference sources (/root/reference)")
@pytest.mark.parametrize("unit,defines", [("exports.cpp", []), ("sd.cpp", ["-DUSE_ONNXSTREAM=1"]), ("llm.cpp", [])])
def test_reference_translation_unit_compiles_against_our_header(unit, defines):
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(os.path.join(REF_SRC, unit), d)
        for h in ("onnxstream.h",):
            shutil.copy(os.path.join(HOST, h), d)
        cmd = ["g++", "-std=c++20", "-fcoroutines", "-fsyntax-only", "-w", *defines, "-I", d, "-I", REF_SRC, "-I", _torch_include(),
               os.path.join(d, unit)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        # the compile must have bound OUR header, not the reference's (same directory as the including file wins)
        dep = subprocess.run(cmd[:-1] + ["-M", os.path.join(d, unit)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    assert r.returncode == 0, r.stdout[-3000:]
    assert os.path.join(d, "onnxstream.h") in dep and os.path.join(REF_SRC, "onnxstream.h") not in dep


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="needs the reference sources (/root/reference)")
def test_our_library_exports_every_symbol_of_the_reference_c_api():
    """every `extern "C"` function src/exports.cpp defines exists in libonnxstream_amd.so (the bindings load them by name)."""
    import re
    from onnxstream_amd import build as b
    if not os.path.exists(b.LIB_HOST):
        pytest.skip("host library not built")
    src = open(os.path.join(REF_SRC, "exports.cpp")).read()
    names = set(re.findall(r"\b(model_[a-z0-9_]+)\s*\(", src))
    have = subprocess.run(["nm", "-D", "--defined-only", b.LIB_HOST], stdout=subprocess.PIPE, text=True).stdout
    missing = sorted(n for n in names if not re.search(r"\b%s\b" % n, have))
    assert names and not missing, missing
