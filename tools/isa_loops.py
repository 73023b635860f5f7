#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel of libosgpu.so (static, no GPU): which loops hold MFMAs, and per loop how many MFMA / LDS reads / LDS-DMA loads /
VALU / SALU / waits / spill moves (v_readlane / v_writelane) they issue.  Used for profiles/r05_isa_loops.txt.

    python tools/isa_loops.py "gemm2_kernel<128, 128, 2, false, 0, 0, 0, 5, 1>" [--valu]

Finds the kernel by (demangled) name substring in every gfx950 code object of the library, disassembles that object with llvm-objdump and walks backward branches.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def functions(lines):
    heads = [(i, l) for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*>:$", l)]
    names = subprocess.run(["c++filt"], input="\n".join(re.search(r"<(.*)>:", l).group(1) for _, l in heads), capture_output=True, text=True).stdout.split("\n")
    for k, ((i, _), n) in enumerate(zip(heads, names)):
        yield n, i, (heads[k + 1][0] if k + 1 < len(heads) else len(lines))


def classify(op, args):
    if op.startswith("v_mfma"):
        return "MFMA " + op
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "LDS read " + op
    if op.startswith("ds_"):
        return "LDS other " + op
    if op.startswith("buffer_load"):
        return "buffer " + op + (" lds" if " lds" in args else "")
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "spill move " + op
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return op


def loops_of(body, valu_detail):
    base = int(re.match(r"^([0-9a-f]+) <", body[0]).group(1), 16)
    ins = []
    for i, l in enumerate(body):
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), l))
    a2i = {a: k for k, (a, _, _, _) in enumerate(ins)}
    found = []
    for k, (a, op, args, l) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", l)
            if m:
                t = base + int(m.group(1), 16)
                if t <= a and t in a2i:
                    found.append((a2i[t], k))
    for s, e in found:
        c = collections.Counter(classify(op, args) for (_, op, args, _) in ins[s:e + 1])
        if not any(k.startswith("MFMA") for k in c):
            continue
        print(f"  loop of {e - s + 1} instructions at +0x{ins[s][0] - base:x} .. +0x{ins[e][0] - base:x}")
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print(f"    {v:5d}  {k}")
        if valu_detail:
            d = collections.Counter(op for (_, op, _, _) in ins[s:e + 1] if op.startswith("v_") and not op.startswith("v_mfma"))
            print("    VALU by opcode: " + ", ".join(f"{v} {k}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])))


def main():
    pat = sys.argv[1]
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "onnxstream_amd", "libosgpu.so")
    blob = open(lib, "rb").read()
    for _, obj in kr.code_objects(blob):
        if not obj.startswith(b"\x7fELF"):
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            lines = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout.split("\n")
        for name, i, e in functions(lines):
            if pat in name:
                print(re.sub(r"\(anonymous namespace\)::", "", name)[:140])
                loops_of(lines[i:e], "--valu" in sys.argv)


if __name__ == "__main__":
    main()
