"""Synthetic Stable Diffusion VAE *decoder* graph (AutoencoderKL, SD 1.5 / SDXL share the topology) in OnnxStream
``model.txt`` format -- the graph the reference app runs after the last denoising step (reference src/sd.cpp:1174-1256
``decoder_solver``: input ``input_2E_1`` [1,4,64,64] = latents * 5.48998, output [1,3,512,512] in [-1,1]).

Topology (diffusers AutoencoderKL decoder): post_quant_conv 1x1 -> conv_in 3x3 (4->C3) -> mid {Resnet, single-head
self-attention over H*W tokens with head dim C3, Resnet} -> 4 up blocks of 3 Resnets (+ nearest 2x upsample + conv on the
first three) -> GroupNorm, SiLU, conv_out (C0->3).  SD: block_out = (128, 256, 512, 512), 32 groups, eps 1e-6.
Op patterns are the exporter's; weights are seeded random like the UNet emitter's.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from .graph import GraphBuilder, T


@dataclass
class VAEConfig:
    block_out: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2          # decoder uses layers_per_block + 1 resnets per up block
    latent_ch: int = 4
    out_ch: int = 3
    latent: int = 64
    groups: int = 32
    in_name: str = "input.1"          # the tiled / SDXL decoders are fed as "latent_sample" (src/sd.cpp:1290)
    name: str = "sd_vae"


SD_VAE = VAEConfig()
TINY_VAE = VAEConfig(block_out=(16, 32, 32), layers_per_block=1, latent=16, groups=8, name="tiny_vae")


class _Decoder:
    def __init__(self, g: GraphBuilder, cfg: VAEConfig):
        self.g, self.cfg = g, cfg

    def resnet(self, name, x: T, cout: int) -> T:
        g, cfg = self.g, self.cfg
        cin = x.shape[1]
        h = g.group_norm(name + "/norm1", x, cfg.groups, 1e-6)
        h = g.silu(name + "/nonlinearity", h)
        h = g.conv(name + "/conv1", h, cout, 3)
        h = g.group_norm(name + "/norm2", h, cfg.groups, 1e-6)
        h = g.silu(name + "/nonlinearity_1", h)
        h = g.conv(name + "/conv2", h, cout, 3)
        if cin != cout:
            x = g.conv(name + "/conv_shortcut", x, cout, 1)
        return g.binary(name + "/Add", "Add", x, h)

    def attention(self, name, x: T) -> T:
        """Single-head spatial self-attention (head dim = C): GroupNorm -> tokens -> q,k,v Linear(+bias) ->
        MatMul . Mul(scale) . Softmax . MatMul (adjacent: the reference fuses it) -> Linear -> back to NCHW -> + residual."""
        g, cfg = self.g, self.cfg
        n, c, h, w = x.shape
        t = h * w
        y = g.group_norm(name + "/group_norm", x, cfg.groups, 1e-6)
        y = g.reshape(name + "/Reshape", y, (1, c, t))
        y = g.transpose(name + "/Transpose", y, (0, 2, 1))                      # [1, T, C]
        q = g.linear(name + "/to_q", y, c)
        k = g.linear(name + "/to_k", y, c)
        v = g.linear(name + "/to_v", y, c)
        kt = g.transpose(name + "/Transpose_1", k, (0, 2, 1))                   # [1, C, T]
        s = g.op(name + "/MatMul", "MatMul", [q, kt], (1, t, t))
        s = g.binary(name + "/Mul", "Mul", s, g.scalar(f"{name}.scale", c ** -0.5))
        p = g.op(name + "/Softmax", "Softmax", [s], s.shape, {"axis": "-1"})
        o = g.op(name + "/MatMul_1", "MatMul", [p, v], (1, t, c))
        o = g.linear(name + "/to_out.0", o, c)
        o = g.transpose(name + "/Transpose_2", o, (0, 2, 1))                    # [1, C, T]
        o = g.reshape(name + "/Reshape_1", o, (1, c, h, w))
        return g.binary(name + "/Add", "Add", o, x)

    def upsample(self, name, x: T) -> T:
        g = self.g
        n, c, h, w = x.shape
        sc = g.weight(f"{name}.scales", np.asarray([1, 1, 2, 2], np.float32), dtype="float32", allow_quant=False, q8_exempt=True)
        r = g.op(name + "/Resize", "Resize", [x, None, sc], (n, c, 2 * h, 2 * w),
                 {"coordinate_transformation_mode": "asymmetric", "cubic_coeff_a": "-0.75", "mode": "nearest",
                  "nearest_mode": "floor"})
        return g.conv(name + "/conv", r, c, 3)

    def build(self):
        g, cfg = self.g, self.cfg
        L = cfg.latent
        z = g.input(cfg.in_name, (1, cfg.latent_ch, L, L))
        x = g.conv("/post_quant_conv", z, cfg.latent_ch, 1)
        cm = cfg.block_out[-1]
        x = g.conv("/decoder/conv_in", x, cm, 3)
        x = self.resnet("/decoder/mid_block/resnets.0", x, cm)
        x = self.attention("/decoder/mid_block/attentions.0", x)
        x = self.resnet("/decoder/mid_block/resnets.1", x, cm)
        rev = list(reversed(cfg.block_out))
        for ui, cout in enumerate(rev):
            for j in range(cfg.layers_per_block + 1):
                x = self.resnet(f"/decoder/up_blocks.{ui}/resnets.{j}", x, cout)
            if ui != len(rev) - 1:
                x = self.upsample(f"/decoder/up_blocks.{ui}/upsamplers.0", x)
        x = g.group_norm("/decoder/conv_norm_out", x, cfg.groups, 1e-6)
        x = g.silu("/decoder/conv_act", x)
        n, c, h, w = x.shape
        wt = g.weight("/decoder/conv_out.weight", g.randn((cfg.out_ch, c, 3, 3), 1.0 / math.sqrt(c * 9)), conv=True)
        b = g.weight("/decoder/conv_out.bias", g.randn((cfg.out_ch,), 0.02), allow_quant=False, q8_exempt=True)
        return g.op("/decoder/conv_out", "Conv", [x, wt, b], (1, cfg.out_ch, h, w),
                    {"dilations": "1,1", "group": "1", "kernel_shape": "3,3", "pads": "1,1,1,1", "strides": "1,1"},
                    out_names=["out_image"])


def build_vae_decoder(sink, cfg: VAEConfig = SD_VAE, wdtype: str = "float16", seed: int = 4321, quant_weights: bool = False,
                      quant_all: bool = False):
    g = GraphBuilder(sink, wdtype=wdtype, seed=seed, quant_weights=quant_weights, quant_all=quant_all)
    out = _Decoder(g, cfg).build()
    g.finish()
    return g, out


def vae_inputs(cfg: VAEConfig, seed: int = 7):
    rng = np.random.default_rng(seed)
    return {cfg.in_name: rng.standard_normal((1, cfg.latent_ch, cfg.latent, cfg.latent), dtype=np.float32)}
