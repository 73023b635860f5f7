#!/usr/bin/env python3
"""Static resource table of every gfx950 kernel in libosgpu.so: VGPRs, AGPRs, SGPRs, LDS, scratch, waves per SIMD the register budget allows.

Reads the offload bundles out of the built library (no GPU), decodes each code object's AMDGPU metadata note with llvm-readelf and prints one line per kernel,
hot kernels (by name filter) first.  The point of the table: no hot kernel spills to scratch, and the waves-per-SIMD figure each tuning discussion in DESIGN.md
quotes can be checked against the binary.

    python tools/kernel_resources.py [--lib onnxstream_amd/libosgpu.so] [--all] > profiles/rNN_kernel_resources.txt
"""
from __future__ import annotations

import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def code_objects(blob: bytes):
    """Yield (triple, bytes) for every device entry of every bundle in the file."""
    at = 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            return
        p = at + len(MAGIC)
        (n,) = struct.unpack_from("<Q", blob, p)
        p += 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tlen].decode()
            p += tlen
            if "gfx" in triple and size:
                yield triple, blob[at + off: at + off + size]
        at = p


def waves_per_simd(vgpr: int, agpr: int) -> int:
    # gfx950: 512 unified registers per lane per SIMD, allocated in blocks of 8.  The metadata's .vgpr_count is the UNIFIED total (architectural VGPRs, rounded up,
    # plus the accumulation registers the compiler used beyond 256); .agpr_count is the part of it that is AGPRs.
    total = (vgpr + 7) // 8 * 8
    return max(1, min(8, 512 // max(total, 1)))


def kernels_of(obj: bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(obj)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, d=0: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, d])[1]
        name = g("name", "?")
        yield dict(name=name, vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")), lds=int(g("group_segment_fixed_size")),
                   scratch=int(g("private_segment_fixed_size")), wg=int(g("max_flat_workgroup_size")), vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")))


def demangle(names):
    out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", re.sub(r"\((osg_mm::GemmParams|AttnParams[^)]*|[^()]*)\)$", "", o)) for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "onnxstream_amd", "libosgpu.so"))
    ap.add_argument("--all", action="store_true", help="every kernel, not only the contraction / attention / normalisation families")
    a = ap.parse_args()
    blob = open(a.lib, "rb").read()
    rows = []
    for triple, obj in code_objects(blob):
        if not obj.startswith(b"\x7fELF"):
            continue
        rows += list(kernels_of(obj))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["pretty"] = n.replace("void ", "")
    hot = re.compile(r"gemm2_kernel|conv3x3_kernel|attn2_kernel|attn_kernel|tblock_tail|gn_|splitk_reduce|qu8|lean|layer_norm")
    rows.sort(key=lambda r: (not hot.search(r["pretty"]), r["pretty"]))
    print(f"# {os.path.basename(a.lib)}: {len(rows)} gfx950 kernels; dynamic LDS (the contraction rings, attention) is not in the static figure")
    print(f"# kernels with scratch: {sum(1 for r in rows if r['scratch'])}; with register spills: {sum(1 for r in rows if r['vspill'] or r['sspill'])}")
    print("# vgpr = unified total (agpr = the part of it that is accumulation registers)")
    print("# vgpr agpr sgpr  lds_static scratch vspill sspill waves/SIMD(regs)  max_wg  kernel")
    for r in rows:
        if not a.all and not hot.search(r["pretty"]):
            continue
        print(f"{r['vgpr']:5d} {r['agpr']:4d} {r['sgpr']:4d} {r['lds']:10d} {r['scratch']:7d} {r['vspill']:6d} {r['sspill']:6d} {waves_per_simd(r['vgpr'], r['agpr']):6d} {r['wg']:12d}  {r['pretty'][:150]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
