"""Synthetic Stable Diffusion UNet graphs (SD 1.5 and SDXL-base topologies) in OnnxStream ``model.txt`` format.

No SD ``model.txt`` exists offline (weights live on Hugging Face, reference src/sd.cpp:3035-3091), so the graph is
generated: the op sequences are those of a diffusers -> torch.onnx.export(opset 14) -> onnx-simplifier -> onnx2txt
export (recipe reference README.md:368-430; per-block patterns SURVEY.md Appendix B) and honour every constraint the
reference's ``Model::run`` enforces.  Weights are seeded random (N(0, 1/fan_in)), inputs are the ones the reference
app pushes: ``timestep``[1], ``sample``[1,4,H,W], ``encoder_hidden_states``[1,77,ctx] (src/sd.cpp:1461-1476).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from .graph import GraphBuilder, T


@dataclass
class UNetConfig:
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # transformer depth per down level (0 = plain Down/UpBlock2D)
    transformer_depth: Tuple[int, ...] = (1, 1, 1, 0)
    mid_depth: int = 1
    heads: int = 8                 # SD1.5: fixed 8 heads; SDXL: head_dim 64 (set head_dim instead)
    head_dim: int = 0              # if >0, heads = C // head_dim
    ctx_dim: int = 768
    ctx_len: int = 77
    in_ch: int = 4
    out_ch: int = 4
    latent: int = 64
    groups: int = 32
    temb_ch: int = 0               # 0 => 4*block_out[0]
    linear_proj: bool = False      # SDXL: linear proj_in/proj_out instead of 1x1 conv
    sdxl_add_embed: bool = False   # SDXL: time_ids + text_embeds additional embedding
    name: str = "sd15"

    def n_heads(self, c):
        return c // self.head_dim if self.head_dim else self.heads


SD15 = UNetConfig()
SDXL = UNetConfig(block_out=(320, 640, 1280), transformer_depth=(0, 2, 10), mid_depth=10, head_dim=64, ctx_dim=2048,
                  latent=128, linear_proj=True, sdxl_add_embed=True, name="sdxl")
# a structurally identical miniature for CPU-side tests (runs through the oracle in well under a second)
TINY = UNetConfig(block_out=(32, 64, 64, 64), heads=2, ctx_dim=48, ctx_len=11, latent=16, name="tiny")
TINY_XL = UNetConfig(block_out=(32, 64, 64), transformer_depth=(0, 1, 2), mid_depth=2, head_dim=16, ctx_dim=48, ctx_len=11,
                     latent=16, linear_proj=True, sdxl_add_embed=True, name="tinyxl")


class _UNet:
    def __init__(self, g: GraphBuilder, cfg: UNetConfig):
        self.g, self.cfg = g, cfg
        self.temb_ch = cfg.temb_ch or 4 * cfg.block_out[0]

    # ---- embeddings ------------------------------------------------------------------------------
    def _sinusoid(self, name, x: T, dim: int) -> T:
        """x:[1,1] (or [N,1]) -> [N,dim] = cos || sin (flip_sin_to_cos, downscale_freq_shift 0)."""
        g = self.g
        half = dim // 2
        freqs = np.exp(-math.log(10000.0) * np.arange(half, dtype=np.float32) / half).reshape(1, half)
        f = g.weight(f"{name}.freqs", freqs, allow_quant=False)
        m = g.binary(name + "/Mul", "Mul", x, f)
        s = g.unary(name + "/Sin", "Sin", m)
        c = g.unary(name + "/Cos", "Cos", m)
        return g.concat(name + "/Concat", [c, s], 1)

    def time_embedding(self, timestep: T, extra=None) -> T:
        g, c0 = self.g, self.cfg.block_out[0]
        t = g.unsqueeze("/time_proj/Unsqueeze", timestep, 1)
        e = self._sinusoid("/time_proj", t, c0)
        h = g.gemm("/time_embedding/linear_1", e, self.temb_ch)
        h = g.silu("/time_embedding/act", h)
        h = g.gemm("/time_embedding/linear_2", h, self.temb_ch)
        if extra is not None:
            h = g.binary("/Add_emb", "Add", h, extra)
        return h

    def add_embedding(self, time_ids: T, text_embeds: T) -> T:
        """SDXL micro-conditioning: time_ids[1,6] -> 6x256 sinusoid -> [1,1536] || text_embeds[1,1280] -> MLP."""
        g = self.g
        n = time_ids.shape[1]
        r = g.reshape("/add_time_proj/Reshape", time_ids, (n, 1))
        e = self._sinusoid("/add_time_proj", r, 256 if self.temb_ch >= 512 else 16)
        e = g.reshape("/add_time_proj/Reshape_1", e, (1, n * e.shape[1]))
        cat = g.concat("/add_embedding/Concat", [text_embeds, e], 1)
        h = g.gemm("/add_embedding/linear_1", cat, self.temb_ch)
        h = g.silu("/add_embedding/act", h)
        return g.gemm("/add_embedding/linear_2", h, self.temb_ch)

    # ---- blocks ----------------------------------------------------------------------------------
    def resnet(self, name, x: T, temb: T, cout: int) -> T:
        g, cfg = self.g, self.cfg
        cin = x.shape[1]
        h = g.group_norm(name + "/norm1", x, cfg.groups, 1e-5)
        h = g.silu(name + "/nonlinearity", h)
        h = g.conv(name + "/conv1", h, cout, 3)
        t = g.silu(name + "/nonlinearity_1", temb)
        t = g.gemm(name + "/time_emb_proj", t, cout)
        t = g.unsqueeze(name + "/Unsqueeze", t, 2)
        t = g.unsqueeze(name + "/Unsqueeze_1", t, 3)
        h = g.binary(name + "/Add", "Add", h, t)
        h = g.group_norm(name + "/norm2", h, cfg.groups, 1e-5)
        h = g.silu(name + "/nonlinearity_2", h)
        h = g.conv(name + "/conv2", h, cout, 3)
        if cin != cout:
            x = g.conv(name + "/conv_shortcut", x, cout, 1)
        return g.binary(name + "/Add_1", "Add", x, h)

    def attention(self, name, x: T, ctx: T) -> T:
        """x:[1,T,C]; ctx:[1,Tk,Cc].  The MatMul/Mul/Softmax/MatMul run is adjacent so the reference fuses it."""
        g, cfg = self.g, self.cfg
        _, tq, c = x.shape
        _, tk, _ = ctx.shape
        h = cfg.n_heads(c)
        d = c // h

        def heads(nm, t: T, tokens):
            r = g.reshape(f"{name}/{nm}/Reshape", t, (1, tokens, h, d))
            p = g.transpose(f"{name}/{nm}/Transpose", r, (0, 2, 1, 3))
            return g.reshape(f"{name}/{nm}/Reshape_1", p, (h, tokens, d))

        q = heads("q", g.matmul_w(name + "/to_q", x, c), tq)
        k = heads("k", g.matmul_w(name + "/to_k", ctx, c), tk)
        v = heads("v", g.matmul_w(name + "/to_v", ctx, c), tk)
        kt = g.transpose(name + "/k/Transpose_1", k, (0, 2, 1))
        s = g.op(name + "/MatMul", "MatMul", [q, kt], (h, tq, tk))
        s = g.binary(name + "/Mul", "Mul", s, g.scalar(f"{name}.scale", d ** -0.5))
        p = g.op(name + "/Softmax", "Softmax", [s], s.shape, {"axis": "-1"})
        o = g.op(name + "/MatMul_1", "MatMul", [p, v], (h, tq, d))
        o = g.reshape(name + "/Reshape_o", o, (1, h, tq, d))
        o = g.transpose(name + "/Transpose_o", o, (0, 2, 1, 3))
        o = g.reshape(name + "/Reshape_o1", o, (1, tq, c))
        return g.linear(name + "/to_out.0", o, c, bias=True)

    def feed_forward(self, name, x: T) -> T:
        g = self.g
        c = x.shape[-1]
        p = g.linear(name + "/net.0/proj", x, 8 * c)
        val = g.slice_last(name + "/net.0/Slice", p, 0, 4 * c)
        gate = g.slice_last(name + "/net.0/Slice_1", p, 4 * c, 8 * c)
        d = g.binary(name + "/net.0/Div", "Div", gate, g.scalar(f"{name}.sqrt2", math.sqrt(2.0)))
        e = g.unary(name + "/net.0/Erf", "Erf", d)
        a = g.binary(name + "/net.0/Add", "Add", e, g.scalar(f"{name}.one", 1.0))
        m = g.binary(name + "/net.0/Mul", "Mul", gate, a)
        m = g.binary(name + "/net.0/Mul_1", "Mul", m, g.scalar(f"{name}.half", 0.5))
        y = g.binary(name + "/net.0/Mul_2", "Mul", val, m)
        return g.linear(name + "/net.2", y, c)

    def basic_transformer(self, name, x: T, ctx: T) -> T:
        g = self.g
        n = g.layer_norm(name + "/norm1", x)
        x = g.binary(name + "/Add", "Add", self.attention(name + "/attn1", n, n), x)
        n = g.layer_norm(name + "/norm2", x)
        x = g.binary(name + "/Add_1", "Add", self.attention(name + "/attn2", n, ctx), x)
        n = g.layer_norm(name + "/norm3", x)
        return g.binary(name + "/Add_2", "Add", self.feed_forward(name + "/ff", n), x)

    def transformer2d(self, name, x: T, ctx: T, depth: int) -> T:
        g, cfg = self.g, self.cfg
        n_, c, h, w = x.shape
        res = x
        y = g.group_norm(name + "/norm", x, cfg.groups, 1e-6)
        if not cfg.linear_proj:
            y = g.conv(name + "/proj_in", y, c, 1)
            y = g.transpose(name + "/Transpose", y, (0, 2, 3, 1))
            y = g.reshape(name + "/Reshape", y, (1, h * w, c))
        else:
            y = g.transpose(name + "/Transpose", y, (0, 2, 3, 1))
            y = g.reshape(name + "/Reshape", y, (1, h * w, c))
            y = g.linear(name + "/proj_in", y, c)
        for i in range(depth):
            y = self.basic_transformer(f"{name}/transformer_blocks.{i}", y, ctx)
        if not cfg.linear_proj:
            y = g.reshape(name + "/Reshape_1", y, (1, h, w, c))
            y = g.transpose(name + "/Transpose_1", y, (0, 3, 1, 2))
            y = g.conv(name + "/proj_out", y, c, 1)
        else:
            y = g.linear(name + "/proj_out", y, c)
            y = g.reshape(name + "/Reshape_1", y, (1, h, w, c))
            y = g.transpose(name + "/Transpose_1", y, (0, 3, 1, 2))
        return g.binary(name + "/Add", "Add", y, res)

    def upsample(self, name, x: T) -> T:
        g = self.g
        n, c, h, w = x.shape
        sc = g.weight(f"{name}.scales", np.asarray([1, 1, 2, 2], np.float32), dtype="float32", allow_quant=False, q8_exempt=True)
        r = g.op(name + "/Resize", "Resize", [x, None, sc], (n, c, 2 * h, 2 * w),
                 {"coordinate_transformation_mode": "asymmetric", "cubic_coeff_a": "-0.75", "mode": "nearest",
                  "nearest_mode": "floor"})
        return g.conv(name + "/conv", r, c, 3)

    # ---- whole net -------------------------------------------------------------------------------
    def build(self):
        g, cfg = self.g, self.cfg
        L = cfg.latent
        sample = g.input("sample", (1, cfg.in_ch, L, L))
        timestep = g.input("timestep", (1,))
        ctx = g.input("encoder_hidden_states", (1, cfg.ctx_len, cfg.ctx_dim))
        extra = None
        if cfg.sdxl_add_embed:
            text_embeds = g.input("text_embeds", (1, 1280 if self.temb_ch >= 512 else 32))
            time_ids = g.input("time_ids", (1, 6))
            extra = self.add_embedding(time_ids, text_embeds)
        temb = self.time_embedding(timestep, extra)

        x = g.conv("/conv_in", sample, cfg.block_out[0], 3)
        skips = [x]
        nlev = len(cfg.block_out)
        for lv, cout in enumerate(cfg.block_out):
            for j in range(cfg.layers_per_block):
                x = self.resnet(f"/down_blocks.{lv}/resnets.{j}", x, temb, cout)
                if cfg.transformer_depth[lv]:
                    x = self.transformer2d(f"/down_blocks.{lv}/attentions.{j}", x, ctx, cfg.transformer_depth[lv])
                skips.append(x)
            if lv != nlev - 1:
                x = g.conv(f"/down_blocks.{lv}/downsamplers.0/conv", x, cout, 3, stride=2, pad=1)
                skips.append(x)

        cm = cfg.block_out[-1]
        x = self.resnet("/mid_block/resnets.0", x, temb, cm)
        if cfg.mid_depth:
            x = self.transformer2d("/mid_block/attentions.0", x, ctx, cfg.mid_depth)
        x = self.resnet("/mid_block/resnets.1", x, temb, cm)

        rev = list(reversed(cfg.block_out))
        rdepth = list(reversed(cfg.transformer_depth))
        for ui, cout in enumerate(rev):
            for j in range(cfg.layers_per_block + 1):
                skip = skips.pop()
                x = g.concat(f"/up_blocks.{ui}/Concat_{j}", [x, skip], 1)
                x = self.resnet(f"/up_blocks.{ui}/resnets.{j}", x, temb, cout)
                if rdepth[ui]:
                    x = self.transformer2d(f"/up_blocks.{ui}/attentions.{j}", x, ctx, rdepth[ui])
            if ui != nlev - 1:
                x = self.upsample(f"/up_blocks.{ui}/upsamplers.0", x)
        assert not skips

        x = g.group_norm("/conv_norm_out", x, cfg.groups, 1e-5)
        x = g.silu("/conv_act", x)
        # last op: give the graph output a stable name
        n, c, h, w = x.shape
        wt = g.weight("/conv_out.weight", g.randn((cfg.out_ch, c, 3, 3), 1.0 / math.sqrt(c * 9)), conv=True)
        b = g.weight("/conv_out.bias", g.randn((cfg.out_ch,), 0.02), allow_quant=False, q8_exempt=True)
        out = g.op("/conv_out", "Conv", [x, wt, b], (1, cfg.out_ch, h, w),
                   {"dilations": "1,1", "group": "1", "kernel_shape": "3,3", "pads": "1,1,1,1", "strides": "1,1"},
                   out_names=["out_sample"])
        return out


def build_unet(sink, cfg: UNetConfig = SD15, wdtype: str = "float16", seed: int = 1234, quant_weights: bool = False):
    """Emit the UNet graph + weights into ``sink``; returns (GraphBuilder, output T)."""
    g = GraphBuilder(sink, wdtype=wdtype, seed=seed, quant_weights=quant_weights)
    out = _UNet(g, cfg).build()
    g.finish()
    return g, out


def unet_inputs(cfg: UNetConfig, seed: int = 42, timestep: float = 999.0):
    """Inputs as the reference app builds them (src/sd.cpp:1456-1476): fp32 host tensors."""
    rng = np.random.default_rng(seed)
    L = cfg.latent
    ins = {
        "timestep": np.asarray([timestep], np.float32),
        "sample": rng.standard_normal((1, cfg.in_ch, L, L), dtype=np.float32),
        "encoder_hidden_states": rng.standard_normal((1, cfg.ctx_len, cfg.ctx_dim), dtype=np.float32),
    }
    if cfg.sdxl_add_embed:
        big = (cfg.temb_ch or 4 * cfg.block_out[0]) >= 512
        ins["text_embeds"] = rng.standard_normal((1, 1280 if big else 32), dtype=np.float32)
        ins["time_ids"] = np.asarray([[1024, 1024, 0, 0, 1024, 1024]], np.float32) if big else \
            np.asarray([[16, 16, 0, 0, 16, 16]], np.float32)
    return ins
