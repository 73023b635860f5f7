"""TEST INFRASTRUCTURE ONLY -- builds a NO-OP stand-in for libosgpu.so so that the HOST logic (graph parsing, fusion passes, lowering, arena
packing, side-stream marks, error paths) can be exercised on a box without a GPU: every entry point of include/osgpu.h exists with its exact
signature; memory calls use the host heap, transfers are memcpy, every compute launch returns success WITHOUT computing anything (outputs stay
zero).  It is reachable only through the OSGPU_LIB environment variable the tests set -- the product never looks for it, and a result
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
    "osg_download": "{ (void)ctx; memcpy(host_dst, src, bytes); return 0; }",
    "osg_copy": "{ (void)ctx; memcpy(dst, src, bytes); return 0; }",
    "osg_memset": "{ (void)ctx; memset(dst, value, bytes); return 0; }",
    "osg_graph_end": "{ (void)ctx; *out = (osg_graph*)calloc(1, 16); return 0; }",
    "osg_graph_destroy": "{ free(g); }",
    "osg_timer_stop": "{ (void)ctx; if (ms) *ms = 0.0f; return 0; }",
    "osg_group_norm_conv3x3_supported": "{ (void)N; (void)H; (void)W; (void)Cin; (void)Cout; return 0; }",
}


def generate() -> str:
    src = open(os.path.join(REPO, "include", "osgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = ['#include <stdlib.h>', '#include <string.h>', '#include "osgpu.h"', ""]
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
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-I", os.path.join(REPO, "include"), c, "-o", so])
    return so


if __name__ == "__main__":
    print(build("/tmp/osgpu_stub"))
