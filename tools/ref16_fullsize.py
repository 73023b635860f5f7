"""Generate tests/golden/ref16_fullsize_<host>.npz from the REFERENCE (oracle/_ref): the reference's own fp16 (and fp32) outputs of the FULL-SIZE nets of
tests/test_fullsize.py on THIS host's CPU -- SD 1.5 UNet, the same as W8A16, the SDXL UNet, the SD VAE decoder -- seed 42, exactly the inputs the tests push.

Why: the reference's fp16 output is not one number per input -- XNNPACK picks its micro-kernels per CPU, and the full-size nets move between
hosts.  The parity rule's third leg was a ratio of the reference's fp16 drift; it is replaced by a PINNED bound: the measured spread between two hosts' reference
outputs (this container's Xeon, the GPU box's EPYC), both committed, max(spread, 1e-3).

usage: ref16_fullsize.py <host tag> [out dir] [cases: sd15,sd15_w8,sdxl,vae]     (needs oracle/_ref; a CPU job: ~1 min SD 1.5, several minutes SDXL)
Stored as float16 (every value of an fp16-arithmetic pass is an f16), the VAE image subsampled 4 x 4; ref32 as float32 (SD 1.5 / SDXL outputs are 64 / 256 KB).
"""
import os
import platform
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.synth import sd_unet, sd_vae          # noqa: E402
from onnxstream_amd.synth.graph import DirSink            # noqa: E402
from oracle import ref as oref                            # noqa: E402


def synth_dir(name, build):
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), name) + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        build(DirSink(d))
        open(d + ".complete", "w").write("ok")
    return d


def cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    tag = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "tests", "golden")
    cases = (sys.argv[3] if len(sys.argv) > 3 else "sd15,sd15_w8,sdxl,vae").split(",")
    assert oref.available(), "oracle/_ref is not built"
    res = {"host": np.array(cpu_name()), "threads": np.array(oref.usable_cores())}
    jobs = {
        "sd15": (lambda: synth_dir("sd15", lambda s: sd_unet.build_unet(s, sd_unet.SD15)), lambda: sd_unet.unet_inputs(sd_unet.SD15, 42), "out_sample"),
        "sd15_w8": (lambda: synth_dir("sd15_w8", lambda s: sd_unet.build_unet(s, sd_unet.SD15, quant_weights=True)), lambda: sd_unet.unet_inputs(sd_unet.SD15, 42), "out_sample"),
        "sdxl": (lambda: synth_dir("sdxl", lambda s: sd_unet.build_unet(s, sd_unet.SDXL)), lambda: sd_unet.unet_inputs(sd_unet.SDXL, 42), "out_sample"),
        "vae": (lambda: synth_dir("sd_vae", lambda s: sd_vae.build_vae_decoder(s, sd_vae.SD_VAE)), lambda: sd_vae.vae_inputs(sd_vae.SD_VAE), "out_image"),
    }
    for c in cases:
        mk, ins, out = jobs[c]
        t0 = time.time()
        d = mk()
        x = ins()
        r16 = oref.run_model(d, x, fp16=True)[out]
        r32 = oref.run_model(d, x, fp16=False)[out]
        assert np.array_equal(r16.astype(np.float16).astype(np.float32), r16), "an fp16-arithmetic output that is not an f16 value"
        if c == "vae":
            r16, r32 = r16[..., ::4, ::4], r32[..., ::4, ::4]
        res[c + "_ref16"] = r16.astype(np.float16)
        res[c + "_ref32"] = r32.astype(np.float32)
        mx = float(np.abs(r32).max())
        print(f"{c}: max|ref32| {mx:.4f}  fp16 drift {float(np.abs(r16 - r32).max()) / mx:.3e}  ({time.time() - t0:.0f} s)", flush=True)
    path = os.path.join(out_dir, f"ref16_fullsize_{tag}.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, os.path.getsize(path), "bytes; host:", res["host"], "threads", int(res["threads"]))


if __name__ == "__main__":
    main()
