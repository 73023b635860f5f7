#!/usr/bin/env python3
"""Static (no GPU): for every kernel of libosgpu.so whose demangled name contains one of the given substrings, the code BYTES a workgroup walks once per launch outside its
main loop -- from the entry to the first MFMA loop (the prologue) and from the end of the last MFMA loop to the end of the function (the epilogue and everything the
compiler laid behind it) -- next to the loop bodies.  Round 6: tools/floor_probe4 shows straight-line code executed once streams into a CU at ~3 ns per instruction
(~14 KiB in 4.4 us from the L2, 5.5 us cold); a launch's prologue and epilogue are exactly that kind of code.
    python tools/isa_phases.py gemm2_kernel conv3x3_kernel attn2_kernel ...        (no arguments: the kernels of the headline plan)
"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr
import isa_loops as il

def main():
    pats = sys.argv[1:] or ["gemm2_kernel", "conv3x3_kernel", "attn2_kernel", "tblock_tail_kernel", "gn_slab_kernel", "splitk_reduce4_kernel", "layer_norm_kernel"]
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "onnxstream_amd", os.environ.get("ISA_LIB", "libosgpu.so"))
    blob = open(lib, "rb").read()
    print("# kernel | total bytes | entry -> first MFMA loop | MFMA loops (bytes each) | last MFMA loop -> end")
    for _, obj in kr.code_objects(blob):
        if not obj.startswith(b"\x7fELF"):
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj); f.flush()
            lines = subprocess.run([il.OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout.split("\n")
        for name, i, e in il.functions(lines):
            if not any(p in name for p in pats):
                continue
            body = lines[i:e]
            base = int(re.match(r"^([0-9a-f]+) <", body[0]).group(1), 16)
            ins = []
            for l in body:
                m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", l)
                if m:
                    ins.append((int(m.group(3), 16) - base, m.group(1), l))
            if not ins:
                continue
            total = ins[-1][0] + 8
            a2i = {a: k for k, (a, _, _) in enumerate(ins)}
            loops = []
            for k, (a, op, l) in enumerate(ins):
                if op.startswith("s_cbranch") or op == "s_branch":
                    m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", l)
                    if m:
                        t = int(m.group(1), 16)
                        if t <= a and t in a2i and any(o.startswith("v_mfma") for (_, o, _) in ins[a2i[t]:k + 1]):
                            loops.append((t, a))
            short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0][:90]
            if not loops:
                print(f"{short:92s} | {total:6d} | (no MFMA loop)")
                continue
            # outermost loops only
            loops.sort()
            outer = []
            for s, t in loops:
                if outer and s >= outer[-1][0] and t <= outer[-1][1]:
                    continue
                if outer and s <= outer[-1][0] and t >= outer[-1][1]:
                    outer[-1] = (s, t); continue
                outer.append((s, t))
            print(f"{short:92s} | {total:6d} | {outer[0][0]:6d} | {' '.join(str(t - s + 8) for s, t in outer):>18s} | {total - outer[-1][1] - 8:6d}")

if __name__ == "__main__":
    main()
