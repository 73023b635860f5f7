"""Generate tests/golden/llama_tiny.npz from the REFERENCE (oracle/_ref): the LLM flow of src/llm.cpp:372-440 on the synthetic tiny Llama of
onnxstream_amd/synth/llama.py -- one prefill of 5 tokens, then 4 greedy decode steps over the growing key/value cache -- with fp16 and with
fp32 arithmetic (m_use_scaled_dp_attn_op off: the oracle's XNNPACK has no SDPA operator, the chain runs through genuine operators)."""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from onnxstream_amd.bindings import Model  # noqa: E402
from onnxstream_amd.synth import llama  # noqa: E402
from onnxstream_amd.synth.graph import DirSink  # noqa: E402
from oracle import ref as oref  # noqa: E402

assert oref.available(), "build the oracle first: make -C oracle ref"
PROMPT = [3, 17, 42, 5, 9]
STEPS = 4


def make(cfg, fname):
  out = {"prompt": np.asarray(PROMPT, np.int64)}
  with tempfile.TemporaryDirectory() as d:
      d += "/"
      llama.build_llama(DirSink(d), cfg)
      toks = None
      # "16": fp16 arithmetic; "32": fp32 arithmetic; "16u": fp16 arithmetic with the layer-norm ops upcast to fp32 (m_requires_upcast, src/llm.cpp:379-383)
      for tag, fp16, up in (("16", True, False), ("32", False, False), ("16u", True, True)):
          m = Model(oref.REF_LIB, 1, "ram+nocache")
          llama.configure(m, cfg, d, ops_cache=fp16, upcast=up)
          logits, past = llama.forward(m, cfg, PROMPT, None, fp16)
          out[f"logits{tag}_0"] = logits
          fed = []
          for s in range(STEPS):
              # both arithmetics are fed the fp16 run's greedy tokens, so that every step compares like with like
              nxt = int(np.argmax(logits[0, -1])) if toks is None else int(toks[s])
              fed.append(nxt)
              logits, past = llama.forward(m, cfg, [nxt], past, fp16)
              out[f"logits{tag}_{s + 1}"] = logits
          if toks is None:
              toks = fed
              out["tokens"] = np.asarray(fed, np.int64)
          for i, p in enumerate(past):
              out[f"past{tag}_{i}"] = p
          m.close()
          print(tag, "tokens", fed, "max|logits|", float(np.abs(logits).max()))
          assert fed == [int(t) for t in toks]
  mx = max(float(np.abs(out[f"logits32_{s}"]).max()) for s in range(STEPS + 1))
  for s in range(STEPS + 1):
      print(f"step {s}: |ref16-ref32|/max = {np.abs(out[f'logits16_{s}'] - out[f'logits32_{s}']).max() / mx:.2e}   "
            f"|ref16u-ref32|/max = {np.abs(out[f'logits16u_{s}'] - out[f'logits32_{s}']).max() / mx:.2e}   "
            f"|ref16u-ref16|/max = {np.abs(out[f'logits16u_{s}'] - out[f'logits16_{s}']).max() / mx:.2e}")
  np.savez_compressed(os.path.join(REPO, "tests", "golden", fname), **out)


make(llama.TINY, "llama_tiny.npz")
make(llama.TINY_WIDE, "llama_tiny_wide.npz")