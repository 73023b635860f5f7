"""Dev tool (GPU box): synthetic UNet through the reference oracle (CPU) and through the HIP backend; prints parity."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.bindings import Model  # noqa: E402
from onnxstream_amd.synth import sd_unet  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402

REF = os.path.join(REPO, "oracle", "_ref", "libonnxstream_ref.so")
OURS = os.path.join(REPO, "onnxstream_amd", "libonnxstream_amd.so")


def run(lib, d, ins, fp16=True, opts=(), n_runs=1, batch=1):
    m = Model(lib, 0, "ram+nocache")
    m.read_file(d + "model.txt")
    outs = []
    for r in range(n_runs):
        for b in range(batch):
            for k, v in ins[b].items():
                m.add_tensor(k, v)
        if r == 0:
            if fp16:
                m.set_use_fp16_arithmetic(True)
            m.set_fuse_ops_in_attention(True)
            for k, v in opts:
                m._set_option(k, v)
        t0 = time.time()
        m.run()
        dt = time.time() - t0
        o, _ = m.get_tensor("out_sample")
        outs.append((o, dt))
        m.clear_tensors()
    return m, outs


def main():
    cfg = getattr(sd_unet, sys.argv[1] if len(sys.argv) > 1 else "TINY")
    d = f"/tmp/synth_{cfg.name}/"
    if not os.path.exists(d + "model.txt"):
        g, _ = sd_unet.build_unet(DirSink(d), cfg)
        print("emitted", len(g.lines), "ops", g.n_params, "params")
    ins = [sd_unet.unet_inputs(cfg, 42), sd_unet.unet_inputs(cfg, 43)]
    have_ref = os.path.exists(REF) and "--noref" not in sys.argv
    if have_ref:
        from oracle import ref as oref
        o32, t32 = oref.run_model(d, ins[0], fp16=False, return_times=True)
        o16, t16 = oref.run_model(d, ins[0], fp16=True, return_times=True, ops_cache=True, runs=2)
        o32, o16 = o32["out_sample"], o16["out_sample"]
        mx = np.abs(o32).max()
        print(f"ref ({oref.usable_cores()} threads): fp32 {t32[0]:.3f}s fp16 {t16} s  |ref16-ref32|/max = {np.abs(o16 - o32).max() / mx:.3e}", flush=True)
    for fusion in (0, 1, 2):
        m, outs = run(OURS, d, ins, opts=(("hip_fusion_level", fusion),), n_runs=4)
        o = outs[0][0]
        ms = m.lib.model_hip_last_pass_ms
        import ctypes
        ms.restype = ctypes.c_double
        ms.argtypes = [ctypes.c_void_p]
        kc = m.lib.model_hip_last_kernel_count
        kc.restype = ctypes.c_ulonglong
        kc.argtypes = [ctypes.c_void_p]
        line = f"ours fusion={fusion}: kernels={kc(m.handle)} pass={ms(m.handle):.3f} ms wall={[round(x[1]*1e3,2) for x in outs]}"
        if have_ref:
            line += f"  |ours-ref16|/max={np.abs(o - o16).max() / mx:.3e} |ours-ref32|/max={np.abs(o - o32).max() / mx:.3e}"
        same = all(np.array_equal(outs[0][0], x[0]) for x in outs[1:])
        print(line, "replay-identical" if same else "REPLAY MISMATCH")
        m.close()
    # batch 2 (cond + uncond in one pass)
    m, outs = run(OURS, d, ins, n_runs=3, batch=2)
    print("batch2 wall", [round(x[1] * 1e3, 2) for x in outs])
    m.close()


if __name__ == "__main__":
    main()
