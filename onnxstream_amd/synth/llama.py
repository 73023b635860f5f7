"""Synthetic Llama-style decoder in OnnxStream's model.txt format, shaped like the reference LLM app's graphs (assets/LLM.md export recipe:
HF LlamaForCausalLM with past_key_values in / out, dynamic axes on the token and cache dimensions; driven by src/llm.cpp:372-440):

    inputs : input_ids [1,T] int64, position_ids [1,T] int64, attention_mask [1,P+T] int64, pkv{2l}, pkv{2l+1} [1,Hkv,P,d] (P may be 0)
    outputs: logits [1,T,V]; opkv{2l}, opkv{2l+1} [1,Hkv,P+T,d] (m_extra_outputs: the caller renames them to pkv* for the next call)

ONE model.txt serves every (T, P): dynamic dimensions are written as 0 (m_support_dynamic_shapes).  Only op forms the reference implements
are used (src/onnxstream.cpp: Gather :6316, Shape :7003, Range :7589, Less :7637, Where :7034, Cast :7352, Expand :7154, ...); the attention
chain is the Transpose/MatMul/Div/Add/Softmax/MatMul form its ScaledDotProductAttention rewrite recognises (:3635-3755).
Random weights (no checkpoints offline)."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from .graph import GraphBuilder, T


@dataclass
class LlamaConfig:
    vocab: int = 96
    hidden: int = 64
    layers: int = 2
    heads: int = 4
    kv_heads: int = 2
    inter: int = 128
    max_pos: int = 64
    eps: float = 1e-5
    rope_theta: float = 10000.0
    name: str = "llama_tiny"

    @property
    def head_dim(self):
        return self.hidden // self.heads


TINY = LlamaConfig()
# Mistral-shaped heads: 128-wide, one key/value head shared by two query heads, a single layer
TINY_WIDE = LlamaConfig(vocab=64, hidden=256, layers=1, heads=2, kv_heads=1, inter=256, max_pos=48, name="llama_tiny_wide")


def build_llama(sink, cfg: LlamaConfig = TINY, seed: int = 777) -> str:
    g = GraphBuilder(sink, seed=seed)
    C, H, Hkv, d, V = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.vocab
    rep = H // Hkv
    ids = g.input("input_ids", (1, 0))
    pos = g.input("position_ids", (1, 0))
    am = g.input("attention_mask", (1, 0))
    pkv = [g.input(f"pkv{i}", (1, Hkv, 0, d)) for i in range(2 * cfg.layers)]

    def i64(name, vals):
        return g.const_i64(name, vals)

    def scalar_i64(name, v):
        return g.weight(name, np.asarray(v, dtype=np.int64).reshape(()), dtype="int64")

    # ---- sizes: T = new tokens, P = cached tokens, S = P + T --------------------------------------------------------------------
    sh_ids = g.op("/Shape", "Shape", [ids], (2,))
    t_s = g.op("/Gather", "Gather", [sh_ids, scalar_i64("idx.one", 1)], [()], {"axis": "0"})[0]
    sh_pkv = g.op("/Shape_1", "Shape", [pkv[0]], (4,))
    p_s = g.op("/Gather_1", "Gather", [sh_pkv, scalar_i64("idx.two", 2)], [()], {"axis": "0"})[0]
    s_s = g.op("/Add", "Add", [t_s, p_s], [()])[0]
    # ---- additive mask [1,1,T,S]: causal over absolute positions (HF _make_causal_mask: arange / Less / masked_fill) + padding term ---------
    kpos = g.op("/Range", "Range", [scalar_i64("rng.zero", 0), s_s, scalar_i64("rng.one", 1)], (0,))
    qpos = g.op("/Range_1", "Range", [p_s, s_s, scalar_i64("rng.one_b", 1)], (0,))
    qpos1 = g.op("/Add_1", "Add", [qpos, scalar_i64("mask.one", 1)], (0,))
    qcol = g.op("/Unsqueeze", "Unsqueeze", [qpos1, i64("mask.axes1", [1])], (0, 1))
    allowed = g.op("/Less", "Less", [kpos, qcol], (0, 0))
    fmin = float(np.finfo(np.float16).min)
    causal = g.op("/Where", "Where", [allowed, g.scalar("mask.zero", 0.0), g.scalar("mask.min", fmin)], (0, 0))
    amf = g.op("/Cast", "Cast", [am], (1, 0), {"to": "1"})
    inv = g.op("/Sub", "Sub", [g.scalar("mask.onef", 1.0), amf], (1, 0))
    padm = g.op("/Mul", "Mul", [inv, g.scalar("mask.min_b", fmin)], (1, 0))
    mask2 = g.op("/Add_2", "Add", [causal, padm], (0, 0))
    mask3 = g.op("/Unsqueeze_1", "Unsqueeze", [mask2, i64("mask.axes0", [0])], (1, 0, 0))
    mask = g.op("/Unsqueeze_2", "Unsqueeze", [mask3, i64("mask.axes0b", [0])], (1, 1, 0, 0))
    # shape of the repeated K / V: [1, Hkv, rep, S, d]
    s_1 = g.op("/Unsqueeze_3", "Unsqueeze", [s_s, i64("shape.axes0", [0])], (1,))
    rep_shape = g.op("/Concat", "Concat", [i64("shape.head", [1, Hkv, rep]), s_1, i64("shape.tail", [d])], (5,), {"axis": "0"})

    # ---- rotary tables (HF LlamaRotaryEmbedding: cos/sin caches gathered by position_ids) ---------------------------------------
    inv_freq = 1.0 / (cfg.rope_theta ** (np.arange(0, d, 2, dtype=np.float64) / d))
    fr = np.outer(np.arange(cfg.max_pos, dtype=np.float64), inv_freq)
    embd = np.concatenate([fr, fr], axis=-1)
    cos_t = g.weight("rotary.cos_cached", np.cos(embd).astype(np.float32), dtype="float32", allow_quant=False)   # (Gather copies the stored type)
    sin_t = g.weight("rotary.sin_cached", np.sin(embd).astype(np.float32), dtype="float32", allow_quant=False)
    cos = g.op("/rotary/Gather", "Gather", [cos_t, pos], (1, 0, d), {"axis": "0"})
    sin = g.op("/rotary/Gather_1", "Gather", [sin_t, pos], (1, 0, d), {"axis": "0"})
    cos = g.op("/rotary/Unsqueeze", "Unsqueeze", [cos, i64("rotary.axes", [1])], (1, 1, 0, d))
    sin = g.op("/rotary/Unsqueeze_1", "Unsqueeze", [sin, i64("rotary.axes_b", [1])], (1, 1, 0, d))

    def rms_norm(name, x):
        p = g.op(name + "/Pow", "Pow", [x, g.scalar(f"{name}.two", 2.0)], (1, 0, C))
        m = g.op(name + "/ReduceMean", "ReduceMean", [p], (1, 0, 1), {"axes": "-1", "keepdims": "1"})
        e = g.op(name + "/Add", "Add", [m, g.scalar(f"{name}.eps", cfg.eps)], (1, 0, 1))
        s = g.op(name + "/Sqrt", "Sqrt", [e], (1, 0, 1))
        r = g.op(name + "/Div", "Div", [g.scalar(f"{name}.one", 1.0), s], (1, 0, 1))
        xn = g.op(name + "/Mul", "Mul", [x, r], (1, 0, C))
        w = g.weight(f"{name}.weight", 1.0 + g.randn((C,), 0.1), allow_quant=False)
        return g.op(name + "/Mul_1", "Mul", [w, xn], (1, 0, C))

    def matmul(name, x, n_out, out_shape):
        k = x.shape[-1]
        w = g.weight(f"{name}.weight", g.randn((k, n_out), 1.0 / math.sqrt(k)))
        return g.op(name + "/MatMul", "MatMul", [x, w], out_shape)

    def split_heads(name, x, heads):
        r = g.op(name + "/Reshape", "Reshape", [x, i64(f"{name}.shape", [1, -1, heads, d])], (1, 0, heads, d), {"allowzero": "0"})
        return g.op(name + "/Transpose", "Transpose", [r], (1, heads, 0, d), {"perm": "0,2,1,3"})

    def rope(name, x, heads):
        half = d // 2

        def sl(nm, a, b):
            return g.op(nm, "Slice", [x, i64(f"{nm}.starts", [a]), i64(f"{nm}.ends", [b]), i64(f"{nm}.axes", [-1]), i64(f"{nm}.steps", [1])],
                        (1, heads, 0, b - a))
        x1, x2 = sl(name + "/Slice", 0, half), sl(name + "/Slice_1", half, d)
        nx2 = g.op(name + "/Neg", "Neg", [x2], (1, heads, 0, half))
        rot = g.op(name + "/Concat", "Concat", [nx2, x1], (1, heads, 0, d), {"axis": "-1"})
        a = g.op(name + "/Mul", "Mul", [x, cos], (1, heads, 0, d))
        b = g.op(name + "/Mul_1", "Mul", [rot, sin], (1, heads, 0, d))
        return g.op(name + "/Add", "Add", [a, b], (1, heads, 0, d))

    def repeat_kv(name, x):
        u = g.op(name + "/Unsqueeze", "Unsqueeze", [x, i64(f"{name}.axes", [2])], (1, Hkv, 1, 0, d))
        e = g.op(name + "/Expand", "Expand", [u, rep_shape], (1, Hkv, rep, 0, d))
        return g.op(name + "/Reshape", "Reshape", [e, i64(f"{name}.shape", [1, H, -1, d])], (1, H, 0, d), {"allowzero": "0"})

    emb_w = g.weight("model.embed_tokens.weight", g.randn((V, C), 1.0), dtype="float32", allow_quant=False)
    x = g.op("/model/embed_tokens/Gather", "Gather", [emb_w, ids], (1, 0, C), {"axis": "0"})
    for l in range(cfg.layers):
        L = f"/model/layers.{l}"
        n = rms_norm(L + "/input_layernorm", x)
        q = rope(L + "/self_attn/rope_q", split_heads(L + "/self_attn/q", matmul(L + "/self_attn/q_proj", n, H * d, (1, 0, H * d)), H), H)
        k = rope(L + "/self_attn/rope_k", split_heads(L + "/self_attn/k", matmul(L + "/self_attn/k_proj", n, Hkv * d, (1, 0, Hkv * d)), Hkv), Hkv)
        v = split_heads(L + "/self_attn/v", matmul(L + "/self_attn/v_proj", n, Hkv * d, (1, 0, Hkv * d)), Hkv)
        k_all = g.op(L + "/self_attn/Concat", "Concat", [pkv[2 * l], k], (1, Hkv, 0, d), {"axis": "2"}, out_names=[f"opkv{2 * l}"])
        v_all = g.op(L + "/self_attn/Concat_1", "Concat", [pkv[2 * l + 1], v], (1, Hkv, 0, d), {"axis": "2"}, out_names=[f"opkv{2 * l + 1}"])
        kr, vr = repeat_kv(L + "/self_attn/repeat_k", k_all), repeat_kv(L + "/self_attn/repeat_v", v_all)
        kt = g.op(L + "/self_attn/Transpose_3", "Transpose", [kr], (1, H, d, 0), {"perm": "0,1,3,2"})
        s = g.op(L + "/self_attn/MatMul", "MatMul", [q, kt], (1, H, 0, 0))
        s = g.op(L + "/self_attn/Div", "Div", [s, g.scalar(f"{L}.sqrt_d", math.sqrt(d))], (1, H, 0, 0))
        s = g.op(L + "/self_attn/Add", "Add", [s, mask], (1, H, 0, 0))
        p = g.op(L + "/self_attn/Softmax", "Softmax", [s], (1, H, 0, 0), {"axis": "-1"})
        o = g.op(L + "/self_attn/MatMul_1", "MatMul", [p, vr], (1, H, 0, d))
        o = g.op(L + "/self_attn/Transpose_4", "Transpose", [o], (1, 0, H, d), {"perm": "0,2,1,3"})
        o = g.op(L + "/self_attn/Reshape_o", "Reshape", [o, i64(f"{L}.o_shape", [1, -1, C])], (1, 0, C), {"allowzero": "0"})
        o = matmul(L + "/self_attn/o_proj", o, C, (1, 0, C))
        x = g.op(L + "/Add", "Add", [x, o], (1, 0, C))
        n = rms_norm(L + "/post_attention_layernorm", x)
        gate = matmul(L + "/mlp/gate_proj", n, cfg.inter, (1, 0, cfg.inter))
        sg = g.op(L + "/mlp/act_fn/Sigmoid", "Sigmoid", [gate], (1, 0, cfg.inter))
        act = g.op(L + "/mlp/act_fn/Mul", "Mul", [gate, sg], (1, 0, cfg.inter))
        up = matmul(L + "/mlp/up_proj", n, cfg.inter, (1, 0, cfg.inter))
        h = g.op(L + "/mlp/Mul", "Mul", [act, up], (1, 0, cfg.inter))
        dn = matmul(L + "/mlp/down_proj", h, C, (1, 0, C))
        x = g.op(L + "/Add_1", "Add", [x, dn], (1, 0, C))
    x = rms_norm("/model/norm", x)
    w = g.weight("lm_head.weight", g.randn((C, V), 1.0 / math.sqrt(C)))
    g.op("/lm_head/MatMul", "MatMul", [x, w], (1, 0, V), out_names=["logits"])
    return g.finish()


def forward(m, cfg: LlamaConfig, input_ids, past, fp16: bool = True):
    """One call of the reference app's `forward` (src/llm.cpp:386-428) through the C API: pushes pkv* (empty on the first call), the int64
    inputs, runs, and returns (logits [1,T,V], new past list).  `m` is a bindings.Model with the options of `configure` set.
    (The reference's C API hands back a float pointer from model_add_tensor, so fp32 inputs must be pushed while fp16 arithmetic is off.)"""
    m.set_use_fp16_arithmetic(False)
    T_new = len(input_ids)
    P = past[0].shape[2] if past is not None else 0
    d, Hkv = cfg.head_dim, cfg.kv_heads
    for i in range(2 * cfg.layers):
        t = past[i] if past is not None else np.zeros((1, Hkv, 0, d), np.float32)
        m.add_tensor(f"pkv{i}", np.ascontiguousarray(t, np.float32))
    m.add_tensor("input_ids", np.asarray([input_ids], np.int64))
    m.add_tensor("position_ids", np.asarray([list(range(P, P + T_new))], np.int64))
    m.add_tensor("attention_mask", np.ones((1, P + T_new), np.int64))
    m.set_use_fp16_arithmetic(fp16)
    m.run()
    logits = m.get_tensor("logits")[0]
    new_past = [m.get_tensor(f"opkv{i}")[0] for i in range(2 * cfg.layers)]
    m.clear_tensors()
    return logits, new_past


UPCAST = ("/input_layernorm/", "/post_attention_layernorm/")   # the ops src/llm.cpp:379-383 runs in fp32 (m_requires_upcast)


def forward_resident(m, cfg: LlamaConfig, input_ids, first: bool, P: int, fp16: bool = True):
    """The same call the way src/llm.cpp:386-428 really makes it: the caches never leave Model::m_data -- m_outputs_convert_set = {"logits"} keeps the
    opkv* outputs in fp16, and the next call renames them to pkv* (:403-407) instead of pushing new tensors.  Returns logits only."""
    T_new = len(input_ids)
    m.set_use_fp16_arithmetic(False)
    if first:
        for i in range(2 * cfg.layers):
            m.add_tensor(f"pkv{i}", np.zeros((1, cfg.kv_heads, 0, cfg.head_dim), np.float32))
    else:
        for i in range(2 * cfg.layers):
            assert m.rename_tensor(f"opkv{i}", f"pkv{i}")
    m.add_tensor("input_ids", np.asarray([input_ids], np.int64))
    m.add_tensor("position_ids", np.asarray([list(range(P, P + T_new))], np.int64))
    m.add_tensor("attention_mask", np.ones((1, P + T_new), np.int64))
    m.set_use_fp16_arithmetic(fp16)
    m.run()
    logits = m.get_tensor("logits")[0]
    m.drop_tensor("logits")          # (get_output moves the result out of m_data, :342-353)
    return logits


def configure(m, cfg: LlamaConfig, model_dir: str, sdpa: bool = False, ops_cache: bool = True, upcast: bool = False):
    """Model options of src/llm.cpp:361-377 that the C API can express (m_requires_upcast is a std::function: not settable from here).
    ops_cache=False for fp32-arithmetic runs of the reference: its ops cache does not survive a second call over fp16-stored weights."""
    m.set_support_dynamic_shapes(True)
    if ops_cache:
        m.set_use_ops_cache(True)
        m.set_use_next_op_cache(True)
    if sdpa:
        m.set_use_scaled_dp_attn_op(True)
    if upcast:
        m.set_upcast_substrings(UPCAST)
    for i in range(2 * cfg.layers):
        m.add_extra_output(f"opkv{i}")
    m.read_file(model_dir + "model.txt")
