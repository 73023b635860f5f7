"""Dev tool (GPU box): does running cond and uncond as two CONCURRENT batch-1 graphs (two contexts, two streams) beat one batch-2
graph?  (per-launch fill/drain of one chain would overlap the other chain's math)"""
import os, sys, threading, time
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

def make(pushes):
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m.read_file(d + "model.txt")
    for r in range(2):
        for i in pushes:
            for k, v in i.items():
                m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True); m.set_fuse_ops_in_attention(True)
        m.run()
        if r == 0: m.clear_tensors()
    return m

N = 30
m2 = make(ins)
m2.hip_replay(5)
t0 = time.perf_counter(); m2.hip_replay(N); t_b2 = (time.perf_counter() - t0) / N
print(f"one graph, batch 2: {t_b2*1e3:.3f} ms/step")
m2.close()
ma, mb = make(ins[:1]), make(ins[1:])
ma.hip_replay(3); mb.hip_replay(3)
t0 = time.perf_counter(); ma.hip_replay(N); t_b1 = (time.perf_counter() - t0) / N
print(f"one graph, batch 1: {t_b1*1e3:.3f} ms/pass")
def work(m): m.hip_replay(N)
ths = [threading.Thread(target=work, args=(m,)) for m in (ma, mb)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
t_2s = (time.perf_counter() - t0) / N
print(f"two concurrent batch-1 graphs: {t_2s*1e3:.3f} ms/step (both passes)")
