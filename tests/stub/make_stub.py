"""TEST INFRASTRUCTURE ONLY -- builds a NO-OP stand-in for libosgpu.so so that the HOST logic (graph parsing, fusion passes, lowering, arena
packing, side-stream marks, error paths) can be exercised on a box without a GPU: every entry point of include/osgpu.h exists with its exact
signature; memory calls use the host heap, transfers are memcpy, the DATA-MOVEMENT entry points (transpose, strided copy, concat, nearest resize,
row gather) and the f16 <-> f32 conversion do what include/osgpu.h says in plain C, every ARITHMETIC launch returns success WITHOUT computing anything
(outputs stay zero).  It is reachable only through the OSGPU_LIB environment variable the tests set -- the product never looks for it, and a result
produced through it is garbage by construction, so it cannot pass for a fallback."""
import os
import re
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SPECIAL = {
    "osg_device_count": "{ return 1; }",
    "osg_init": "{ (void)device; *out = (osg_ctx*)calloc(1, 64); return *out ? 0 : 1; }",
    "osg_destroy": "{ free(ctx); }",
    "osg_last_error": '{ (void)ctx; return "stub"; }',
    "osg_device_name": '{ (void)ctx; return "cpu-stub (no compute)"; }',
    "osg_stream": "{ (void)ctx; return 0; }",
    "osg_malloc": "{ (void)ctx; *dptr = calloc(1, bytes ? bytes : 1); return *dptr ? 0 : 1; }",
    "osg_free": "{ (void)ctx; free(dptr); return 0; }",
    "osg_upload": "{ (void)ctx; memcpy(dst, host_src, bytes); return 0; }",
    "osg_upload_sync": "{ (void)ctx; memcpy(dst, host_src, bytes); return 0; }",
    "osg_upload_pinned": "{ (void)ctx; memcpy(dst, pinned_host_src, bytes); return 0; }",
    "osg_upload_pinned_async": "{ (void)ctx; memcpy(dst, pinned_host_src, bytes); return 0; }",
    "osg_download": "{ (void)ctx; memcpy(host_dst, src, bytes); return 0; }",
    "osg_copy": "{ (void)ctx; memcpy(dst, src, bytes); return 0; }",
    "osg_memset": "{ (void)ctx; memset(dst, value, bytes); return 0; }",
    "osg_graph_end": "{ (void)ctx; *out = (osg_graph*)calloc(1, 16); return 0; }",
    "osg_graph_destroy": "{ free(g); }",
    "osg_timer_stop": "{ (void)ctx; if (ms) *ms = 0.0f; return 0; }",
    # (shape predicates and sizes the PLANNER branches on: the real library's answers, so the CPU tests see the plan a GPU box would build)
    "osg_tblock_tail_supported": "{ return C == 320 && heads == 8 && M > 0 && M % 32 == 0 && rows_per_img % 32 == 0 && Tk >= 1 && Tk <= 80; }",
    "osg_tblock_kv_pack_elems": "{ return (size_t)imgs * heads * 80 * (size_t)((D + 15) / 16 * 16); }",
    # ---- DATA MOVEMENT and dtype conversion are real (plain C restatements of the entry points' documented semantics, include/osgpu.h): a graph made
    # of zero-FLOP ops then carries real values end to end on a CPU, which lets tests/test_movement_cpu.py check what the PLANNER hands these entry
    # points -- shapes, permutations, pitches, offsets, layouts -- against numpy.  No arithmetic kernel is implemented: their outputs stay zero.
    "osg_transpose": """{
    (void)ctx;
    long os[8], st[8], ost[8], idx[8] = {0}, n = 1;
    if (rank < 1 || rank > 8) return 1;
    for (int i = rank - 1, acc = 1; i >= 0; i--) { st[i] = acc; acc *= (int)shape[i]; }
    { long acc = 1; for (int i = rank - 1; i >= 0; i--) { st[i] = acc; acc *= shape[i]; } }
    for (int i = 0; i < rank; i++) { os[i] = shape[perm[i]]; ost[i] = st[perm[i]]; n *= os[i]; }
    const char* src = (const char*)x; char* dst = (char*)y;
    for (long o = 0; o < n; o++) {
        long off = 0;
        for (int i = 0; i < rank; i++) off += idx[i] * ost[i];
        memcpy(dst + o * elem_size, src + off * elem_size, (size_t)elem_size);
        for (int i = rank - 1; i >= 0; i--) { if (++idx[i] < os[i]) break; idx[i] = 0; }
    }
    return 0;
}""",
    "osg_copy_2d": """{
    (void)ctx;
    for (long o = 0; o < outer; o++)
        memcpy((char*)dst + (o * dst_pitch + dst_off) * elem_size, (const char*)src + (o * src_pitch + src_off) * elem_size, (size_t)(inner * elem_size));
    return 0;
}""",
    "osg_concat2": """{
    (void)ctx;
    for (long o = 0; o < outer; o++) {
        char* d = (char*)dst + o * (inner_a + inner_b) * elem_size;
        memcpy(d, (const char*)a + o * inner_a * elem_size, (size_t)(inner_a * elem_size));
        memcpy(d + inner_a * elem_size, (const char*)b + o * inner_b * elem_size, (size_t)(inner_b * elem_size));
    }
    return 0;
}""",
    "osg_resize_nearest": """{
    (void)ctx;
    const float shi = (float)H / (float)Ho, swi = (float)W / (float)Wo;
    for (int b = 0; b < N; b++)
        for (int c = 0; c < C; c++)
            for (int ho = 0; ho < Ho; ho++)
                for (int wo = 0; wo < Wo; wo++) {
                    int hi = (int)floorf((float)ho * shi), wi = (int)floorf((float)wo * swi);
                    if (hi > H - 1) hi = H - 1;
                    if (wi > W - 1) wi = W - 1;
                    const long s = nhwc ? (((long)b * H + hi) * W + wi) * C + c : (((long)b * C + c) * H + hi) * W + wi;
                    const long d = nhwc ? (((long)b * Ho + ho) * Wo + wo) * C + c : (((long)b * C + c) * Ho + ho) * Wo + wo;
                    memcpy((char*)y + d * elem_size, (const char*)x + s * elem_size, (size_t)elem_size);
                }
    return 0;
}""",
    "osg_gather_rows": """{
    (void)ctx;
    for (long i = 0; i < n_idx; i++) {
        if (idx[i] < 0 || idx[i] >= n_rows) return 1;
        memcpy((char*)y + i * row_elems * elem_size, (const char*)x + idx[i] * row_elems * elem_size, (size_t)(row_elems * elem_size));
    }
    return 0;
}""",
    "osg_convert": """{
    (void)ctx; (void)scale; (void)zero_point;
    if (src_dtype == dst_dtype) { memcpy(y, x, (size_t)n * (src_dtype == OSG_F32 ? 4 : src_dtype == OSG_F16 ? 2 : src_dtype == OSG_I64 ? 8 : 1)); return 0; }
    if (src_dtype == OSG_F32 && dst_dtype == OSG_F16) { for (long i = 0; i < n; i++) ((unsigned short*)y)[i] = stub_f2h(((const float*)x)[i]); return 0; }
    if (src_dtype == OSG_F16 && dst_dtype == OSG_F32) { for (long i = 0; i < n; i++) ((float*)y)[i] = stub_h2f(((const unsigned short*)x)[i]); return 0; }
    return 0;   /* (uint8 conversions are arithmetic of the uint8 path: not implemented here) */
}""",
}


HALF_HELPERS = r"""
/* IEEE binary16 <-> binary32, round to nearest even (what v_cvt_f16_f32 / the reference's fp16 conversion do) */
static unsigned short stub_f2h(float f) {
    unsigned int x; memcpy(&x, &f, 4);
    const unsigned int sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);              /* rounds to infinity */
    if (x < 0x33000001u) return (unsigned short)sign;                            /* rounds to zero */
    int e = (int)(x >> 23) - 127;
    unsigned int m = (x & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    unsigned int h = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    if (e >= -14) h += (unsigned int)(e + 14) << 10;                             /* (h holds the implicit bit: it carries into the exponent) */
    return (unsigned short)(sign | h);
}
static float stub_h2f(unsigned short h) {
    const unsigned int sign = (unsigned int)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    unsigned int x;
    if (e == 31u) x = sign | 0x7f800000u | (m << 13);
    else if (e) x = sign | ((e + 112u) << 23) | (m << 13);
    else if (!m) x = sign;
    else { float f = (float)m * 5.9604644775390625e-08f; memcpy(&x, &f, 4); x |= sign; }
    float f; memcpy(&f, &x, 4); return f;
}
"""


def generate() -> str:
    src = open(os.path.join(REPO, "include", "osgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = ['#include <math.h>', '#include <stdlib.h>', '#include <string.h>', '#include "osgpu.h"', "", HALF_HELPERS, ""]
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(osg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.M | re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        if name in SPECIAL:
            body.append(f"{ret} {name}({params}) {SPECIAL[name]}")
        elif ret == "int":
            body.append(f"{ret} {name}({params}) {{ return 0; }}")
        elif ret == "void":
            body.append(f"{ret} {name}({params}) {{ }}")
        else:
            raise RuntimeError(f"stub: no rule for '{ret} {name}'")
    return "\n".join(body) + "\n"


def build(out_dir: str) -> str:
    os.makedirs(out_dir, exist_ok=True)
    c = os.path.join(out_dir, "osgpu_stub.c")
    so = os.path.join(out_dir, "libosgpu_stub.so")
    open(c, "w").write(generate())
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-I", os.path.join(REPO, "include"), c, "-o", so, "-lm"])
    return so


if __name__ == "__main__":
    print(build("/tmp/osgpu_stub"))
