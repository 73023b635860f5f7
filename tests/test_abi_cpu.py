"""CPU-side checks of the drop-in boundary (no GPU needed, no compute calls):
  * libosgpu.so loads and exports EVERY entry point include/osgpu.h declares (header parsed, not a hand-kept list);
  * libonnxstream_amd.so exports the reference's model_* C API (reference src/exports.cpp:42-311) and the host logic behind it
    (model.txt parser, tensor push/read-back, name mangling) behaves like the reference's;
  * without a GPU the product path FAILS LOUDLY -- there is no CPU fallback to fall into.
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(REPO, "include", "osgpu.h")


@pytest.fixture(scope="module")
def libs():
    from onnxstream_amd import build as b
    if not (os.path.exists(b.LIB_GPU) and os.path.exists(b.LIB_HOST)):
        import __graft_entry__ as ge
        ge.build()
    return b.LIB_GPU, b.LIB_HOST


def _declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(osg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_surface():
    names = _declared()
    assert len(names) >= 35
    for must in ("osg_init", "osg_conv2d_nhwc", "osg_gemm", "osg_attention", "osg_group_norm_nhwc", "osg_layer_norm", "osg_geglu",
                 "osg_convert", "osg_upload", "osg_download", "osg_graph_begin"):
        assert must in names


def test_libosgpu_exports_every_declared_symbol(libs):
    lib = ctypes.CDLL(libs[0])            # loads without a GPU (HIP runtime is only touched by osg_init)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/osgpu.h but not exported by libosgpu.so"
    from onnxstream_amd import osgpu
    assert sorted(osgpu.EXPORTS) == _declared()   # the ctypes binding covers the same surface


def test_host_library_exports_reference_c_api(libs):
    host = ctypes.CDLL(libs[1])
    for name in ("model_new", "model_new_2", "model_delete", "model_read_file", "model_read_string", "model_get_weights_names",
                 "model_add_weights_file", "model_add_tensor", "model_get_tensor", "model_get_all_tensor_names", "model_run",
                 "model_run_2", "model_clear_tensors", "model_set_option", "model_add_extra_output", "model_free_buffer"):
        assert hasattr(host, name), name


def test_host_logic_parser_and_tensors(libs):
    from onnxstream_amd.bindings import Model
    m = Model(libs[1], 0, "ram")
    m.read_string("c1:Conv*input:x(1,4,8,8);w_nchw.bin(float16:8,4,3,3);b.bin(float16:8)*output:y(1,8,8,8)*"
                  "dilations:1,1;group:1;kernel_shape:3,3;pads:1,1,1,1;strides:1,1\n")
    x = np.arange(256, dtype=np.float32).reshape(1, 4, 8, 8)
    m.add_tensor("x", x)
    got, shape = m.get_tensor("x")
    assert shape == [1, 4, 8, 8] and np.array_equal(got, x)
    assert m.get_all_tensor_names() == ["x"]
    m.clear_tensors()
    assert m.get_all_tensor_names() == []
    assert Model.mangle_name("a/b.c") == "a_2F_b_2E_c" and Model.demangle_name("a_2F_b_2E_c") == "a/b.c"
    m.close()


def test_no_gpu_means_loud_failure(libs):
    """On a box without a GPU, run() must raise (never silently compute on the CPU).  Run in a subprocess so a GPU box, where the
    run would SUCCEED past backend creation, is handled too: there the missing weight file makes it fail instead."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from onnxstream_amd.bindings import Model, OnnxStreamError\n"
        "import numpy as np\n"
        "m = Model(%r, 0, 'ram')\n"
        "m.read_string('a1:Add*input:x(1,4);x(1,4)*output:y(1,4)\\n')\n"
        "m.add_tensor('x', np.ones((1,4), np.float32))\n"
        "m.set_use_fp16_arithmetic(True)\n"
        "try:\n"
        "    m.run(); print('RAN')\n"
        "except OnnxStreamError as e:\n"
        "    print('RAISED', e)\n" % (REPO, libs[1]))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "RAISED" in r.stdout and "no CPU fallback" in r.stdout, (r.stdout, r.stderr[-500:])
