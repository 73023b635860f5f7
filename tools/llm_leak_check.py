import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import llama
from onnxstream_amd.synth.graph import DirSink
cfg = llama.LlamaConfig(vocab=512, hidden=256, layers=4, heads=8, kv_heads=2, inter=512, max_pos=1024, name="leak")
d = tempfile.mkdtemp() + "/"
llama.build_llama(DirSink(d), cfg)
m = Model(b.LIB_HOST, 0, "ram+nocache")
m.add_outputs_convert("logits")
llama.configure(m, cfg, d, sdpa=True, upcast=True)
lg = llama.forward_resident(m, cfg, [1, 2, 3, 4], True, 0)
P = 4
free0 = None
for k in range(600):
    lg = llama.forward_resident(m, cfg, [int(np.argmax(lg[0, -1]))], False, P); P += 1
    if k in (50, 300, 599):
        torch.cuda.synchronize()
        f, t = torch.cuda.mem_get_info()
        print(k, "free MB", f / 1e6, flush=True)
assert np.isfinite(lg).all()
