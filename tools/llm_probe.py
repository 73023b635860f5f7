"""Dev tool (GPU box): the LLM flow (synth/llama.py, src/llm.cpp's loop shape) at TinyLlama-1.1B size -- 22 layers, hidden 2048, 32 query / 4 key-value
heads of 64, MLP 5632, vocabulary 32000, random weights -- prefill of 32 tokens, then greedy decode; wall ms per call, caches resident in the Model (fp16, renamed)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onnxstream_amd import build as b
from onnxstream_amd.bindings import Model
from onnxstream_amd.synth import llama
from onnxstream_amd.synth.graph import DirSink

cfg = llama.LlamaConfig(vocab=32000, hidden=2048, layers=22, heads=32, kv_heads=4, inter=5632, max_pos=2048, name="tinyllama_1b")
d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), cfg.name) + "/"
if not os.path.exists(d + ".complete"):
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    llama.build_llama(DirSink(d), cfg)
    open(d + ".complete", "w").write("ok")
    print(f"emitted {cfg.name} in {time.time()-t0:.1f} s", flush=True)
for sdpa in (True, False):
    m = Model(b.LIB_HOST, 0, "ram+nocache")
    m._set_option("hip_autotune", 0)
    m.add_outputs_convert("logits")
    if os.environ.get("LLM_RESIDENT_OUTPUTS") == "1":      # the caches never leave the device (Model::m_hip_resident_outputs)
        m._set_option("hip_resident_outputs", 1)
    llama.configure(m, cfg, d, sdpa=sdpa, upcast=True)
    rng = np.random.default_rng(0)
    prompt = [int(t) for t in rng.integers(0, cfg.vocab, 32)]
    t0 = time.perf_counter()
    lg = llama.forward_resident(m, cfg, prompt, True, 0)
    t1 = time.perf_counter()
    print(f"sdpa={sdpa}: first call (weights become resident, prefill 32 tokens): {(t1-t0)*1e3:.0f} ms, plan {m.hip_last_kernel_count()} launches, device {m.hip_last_pass_ms():.2f} ms", flush=True)
    times, dev = [], []
    P = len(prompt)
    for k in range(24):
        nxt = int(np.argmax(lg[0, -1]))
        t0 = time.perf_counter()
        lg = llama.forward_resident(m, cfg, [nxt], False, P)
        times.append((time.perf_counter() - t0) * 1e3)
        dev.append(m.hip_last_pass_ms())
        P += 1
    assert np.isfinite(lg).all()
    print(f"sdpa={sdpa}: decode wall ms/token median {np.median(times):.1f} (min {min(times):.1f}), device ms/token median {np.median(dev):.2f}, launches {m.hip_last_kernel_count()}", flush=True)
    m.close()
