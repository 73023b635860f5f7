"""Dev tool (GPU box): the streamed-weights mode on the full SD 1.5 UNet -- every pass re-pulls the 1.72 GB of weights through the
WeightsProvider and streams them H2D against compute.  Reports ms per pass and GB/s against the PCIe Gen5 x16 figure (63 GB/s)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import sd_unet
from onnxstream_amd.synth.graph import DirSink
cfg = sd_unet.SD15
d = "/tmp/onnxstream_amd_synth/sd15/"
if not os.path.exists(d + ".complete"):
    sd_unet.build_unet(DirSink(d), cfg); open(d + ".complete", "w").write("ok")
ins = [sd_unet.unet_inputs(cfg, 42), sd_unet.unet_inputs(cfg, 43)]
for wp in sys.argv[1:] or ["ram+nocache", "nocache"]:
    m = Model(b.LIB_HOST, 0, wp)
    m.read_file(d + "model.txt")
    m._set_option("hip_stream_weights", 1)
    for r in range(4):
        for i in ins:
            for k, v in i.items():
                m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
        t0 = time.perf_counter(); m.run(); dt = time.perf_counter() - t0
        gb = m.hip_streamed_bytes() / 1e9
        print(f"[{wp}] pass {r}: wall {dt*1e3:8.1f} ms  device {m.hip_last_pass_ms():8.1f} ms  streamed {gb:.3f} GB -> {gb/dt if gb else 0:.1f} GB/s of 63 (PCIe Gen5 x16)", flush=True)
        m.clear_tensors()
    m.close()
