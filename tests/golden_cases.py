"""Hot-path golden cases: small graphs, each one a pattern of the SD UNet as the reference's exporter writes it.

The reference ships no tests and no expected outputs, so the golden vectors are OUTPUTS OF THE REFERENCE
ITSELF: tools/make_golden.py runs every case through oracle/_ref (the unmodified /root/reference sources + XNNPACK) in fp16
and fp32 arithmetic and commits inputs + outputs as tests/golden/<case>.npz.  Graph + weights are NOT stored: they are
re-emitted deterministically from the seeds below (numpy Generator streams are stable), so fixtures stay a few KB each.

Used by: tools/make_golden.py (generator), tests/test_golden.py (CPU: oracle/_ref and the numpy restatement against the
fixtures; GPU: the HIP backend against the fixtures -- works where /root/reference does not exist).
"""
from __future__ import annotations

import numpy as np

from onnxstream_amd.synth import sd_unet, sd_vae
from onnxstream_amd.synth.graph import GraphBuilder

f32 = np.float32


def _rn(seed, shape, std=1.0):
    return (np.random.default_rng(seed).standard_normal(shape, dtype=f32) * f32(std)).astype(f32)


def _unet(g, **kw):
    cfg = sd_unet.UNetConfig(block_out=(32, 64), heads=2, ctx_dim=48, ctx_len=11, latent=8, groups=8, name="case", **kw)
    return sd_unet._UNet(g, cfg)


# every builder: (GraphBuilder) -> dict of fp32 inputs.  The graph output is the single unconsumed tensor.
def conv3x3(g):
    x = g.input("x", (1, 64, 12, 12))
    g.conv("/c", x, 128, 3)
    return {"x": _rn(1, (1, 64, 12, 12))}


def conv3x3_stride2(g):
    x = g.input("x", (1, 64, 12, 12))
    g.conv("/c", x, 64, 3, stride=2, pad=1)
    return {"x": _rn(2, (1, 64, 12, 12))}


def conv1x1_nobias(g):
    x = g.input("x", (1, 64, 8, 8))
    g.conv("/c", x, 32, 1, bias=False)
    return {"x": _rn(3, (1, 64, 8, 8))}


def conv_in_4ch(g):
    x = g.input("x", (1, 4, 16, 16))
    g.conv("/c", x, 32, 3)
    return {"x": _rn(4, (1, 4, 16, 16))}


def conv_ragged(g):   # Cin not a multiple of 8, odd spatial size, Cout not a multiple of 4
    x = g.input("x", (1, 20, 7, 5))
    g.conv("/c", x, 30, 3)
    return {"x": _rn(5, (1, 20, 7, 5))}


def linear_bias(g):
    x = g.input("x", (1, 77, 64))
    g.linear("/l", x, 128)
    return {"x": _rn(6, (1, 77, 64))}


def gemm_temb(g):
    x = g.input("x", (1, 128))
    g.gemm("/g", g.silu("/act", x), 64)
    return {"x": _rn(7, (1, 128))}


def group_norm_silu(g):
    x = g.input("x", (1, 64, 8, 8))
    g.silu("/act", g.group_norm("/gn", x, 8, 1e-5))
    return {"x": _rn(8, (1, 64, 8, 8), 2.0) + f32(0.5)}


def layer_norm(g):
    x = g.input("x", (1, 64, 96))
    g.layer_norm("/ln", x)
    return {"x": _rn(9, (1, 64, 96), 1.5) - f32(0.3)}


def self_attention(g):
    x = g.input("x", (1, 64, 64))
    _unet(g).attention("/attn1", x, x)
    return {"x": _rn(10, (1, 64, 64))}


def cross_attention(g):
    x = g.input("x", (1, 64, 64))
    c = g.input("ctx", (1, 11, 48))
    _unet(g).attention("/attn2", x, c)
    return {"x": _rn(11, (1, 64, 64)), "ctx": _rn(12, (1, 11, 48))}


def geglu_ff(g):
    x = g.input("x", (1, 64, 32))
    _unet(g).feed_forward("/ff", x)
    return {"x": _rn(13, (1, 64, 32))}


def resnet_block(g):
    x = g.input("x", (1, 32, 8, 8))
    t = g.input("temb", (1, 128))
    _unet(g).resnet("/res", x, t, 64)
    return {"x": _rn(14, (1, 32, 8, 8)), "temb": _rn(15, (1, 128))}


def transformer_block(g):
    x = g.input("x", (1, 64, 8, 8))
    c = g.input("ctx", (1, 11, 48))
    _unet(g).transformer2d("/tr", x, c, 1)
    return {"x": _rn(16, (1, 64, 8, 8)), "ctx": _rn(17, (1, 11, 48))}


def transformer_block_320(g):
    """the SD 1.5 64x64-level spatial transformer at its real widths (320 channels, 8 heads of 40, text context 77 x 768, 32 groups) on an 8 x 8 image:
    GroupNorm -> proj_in -> BasicTransformerBlock -> proj_out + residual.  The shape osg_tblock_tail takes: at fusion level 2 everything behind the
    self-attention of this graph is ONE launch (round 4)."""
    cfg = sd_unet.UNetConfig(block_out=(320, 640), heads=8, ctx_dim=768, ctx_len=77, latent=8, groups=32, name="case320")
    x = g.input("x", (1, 320, 8, 8))
    c = g.input("ctx", (1, 77, 768))
    sd_unet._UNet(g, cfg).transformer2d("/tr", x, c, 1)
    return {"x": _rn(21, (1, 320, 8, 8)), "ctx": _rn(22, (1, 77, 768))}


def transformer_block_640(g):
    """the same at the 32x32 level's widths (640 channels, 8 heads of 80): LayerNorm folded into attn2.to_q, then the cross-attention launch, at fusion level 2"""
    cfg = sd_unet.UNetConfig(block_out=(640, 1280), heads=8, ctx_dim=768, ctx_len=77, latent=8, groups=32, name="case640")
    x = g.input("x", (1, 640, 8, 8))
    c = g.input("ctx", (1, 77, 768))
    sd_unet._UNet(g, cfg).transformer2d("/tr", x, c, 1)
    return {"x": _rn(23, (1, 640, 8, 8)), "ctx": _rn(24, (1, 77, 768))}


def transformer_block_1280(g):
    """... and at the 16x16 / 8x8 levels' widths (1280 channels, 8 heads of 160): the 160-wide heads"""
    cfg = sd_unet.UNetConfig(block_out=(1280, 1280), heads=8, ctx_dim=768, ctx_len=77, latent=8, groups=32, name="case1280")
    x = g.input("x", (1, 1280, 8, 8))
    c = g.input("ctx", (1, 77, 768))
    sd_unet._UNet(g, cfg).transformer2d("/tr", x, c, 1)
    return {"x": _rn(25, (1, 1280, 8, 8)), "ctx": _rn(26, (1, 77, 768))}


def upsample_concat(g):
    x = g.input("x", (1, 32, 6, 6))
    s = g.input("skip", (1, 32, 12, 12))
    u = _unet(g).upsample("/up", x)
    cat = g.concat("/cat", [u, s], 1)
    g.conv("/c", cat, 32, 3)
    return {"x": _rn(18, (1, 32, 6, 6)), "skip": _rn(19, (1, 32, 12, 12))}


def time_embedding(g):
    t = g.input("timestep", (1,))
    _unet(g).time_embedding(t)
    return {"timestep": np.asarray([999.0], f32)}


def shape_gather_chain(g):
    """what exporters leave around a `view(b, -1, h, d)`: Shape -> Gather (0-d and 1-d indices) -> Unsqueeze -> Cast -> Concat -> Reshape
    (folded to constants at plan time on the device backend), plus an embedding-style Gather of weight rows (reference :6316, :7003, :7352)"""
    x = g.input("x", (1, 6, 32))
    tab = g.weight("/emb.weight", g.randn((10, 32), 1.0))
    idx = g.weight("/emb.idx", np.asarray([[3, 0, 9, 9, 1, 7]], np.int64), dtype="int64")
    e = g.op("/emb/Gather", "Gather", [tab, idx], (1, 6, 32), {"axis": "0"})
    y = g.binary("/add", "Add", x, e)
    sh = g.op("/Shape", "Shape", [y], (3,))
    b0 = g.op("/Gather_b", "Gather", [sh, g.weight("/c0", np.asarray(0, np.int64), dtype="int64")], [()], {"axis": "0"})[0]
    b = g.unsqueeze("/Unsq_b", b0, 0)
    t = g.op("/Gather_t", "Gather", [sh, g.weight("/c1", np.asarray(1, np.int64), dtype="int64")], [()], {"axis": "0"})[0]
    t1 = g.unsqueeze("/Unsq_t", t, 0)
    c = g.op("/Cast", "Cast", [t1], (1,), {"to": "7"})
    tgt = g.concat("/Concat", [b, c, g.const_i64("/hd", [4, 8])], 0)
    r = g.op("/Reshape", "Reshape", [y, tgt], (1, 6, 4, 8), {"allowzero": "0"})
    g.transpose("/T", r, (0, 2, 1, 3))
    return {"x": _rn(20, (1, 6, 32))}


def conv1d_pair(g):   # Conv1D lifted to 2-D (reference src/onnxstream.cpp:4521-4544): stride 1, SiLU, stride 2 -- 64 channels (the direct-to-LDS kernel) and a ragged count
    # (an op between the two: the reference cannot feed a Conv1D's 3-D NHWC result straight into another Conv -- :2917-2920 lifts unspecified-layout inputs only)
    x = g.input("x", (1, 64, 40))
    h = g.silu("/act", g.conv1d("/c1", x, 24, 3))
    g.silu("/act2", g.conv1d("/c2", h, 32, 3, stride=2))   # (nor hand a 3-D NHWC tensor out as a graph output, :8250-8253)
    return {"x": _rn(21, (1, 64, 40))}


CASES = [conv3x3, conv3x3_stride2, conv1x1_nobias, conv_in_4ch, conv_ragged, linear_bias, gemm_temb, group_norm_silu, layer_norm,
         self_attention, cross_attention, geglu_ff, resnet_block, transformer_block, upsample_concat, time_embedding, shape_gather_chain, conv1d_pair]
# longer chains at real widths (15 rounding points deep): held to the whole-net bound of tests/test_golden.py, not the single-pattern one
CHAINS = [transformer_block_320, transformer_block_640, transformer_block_1280]
# whole (miniature) networks: SD1.5-shaped and SDXL-shaped UNets, the VAE decoder (single 32-wide attention head + 3 resolutions)
UNETS = {"unet_tiny": sd_unet.TINY, "unet_tinyxl": sd_unet.TINY_XL, "vae_tiny": sd_vae.TINY_VAE,
         # W8A16 (BASELINE config 3, UNet half): uint8 weights + per-tensor scale/zero-point in model.txt, dequantised at load
         # (reference get_tensor_data :2887-2891 -> Model::dequantize :3353)
         "unet_tiny_w8": (sd_unet.TINY, True)}


def emit(case, sink, seed=1234):
    """Emit a case's graph + weights into `sink`; returns the fp32 input dict."""
    if isinstance(case, str):
        cfg = UNETS[case]
        if isinstance(cfg, tuple):
            cfg, quant = cfg
            sd_unet.build_unet(sink, cfg, seed=seed, quant_weights=quant)
            return sd_unet.unet_inputs(cfg, 42)
        if isinstance(cfg, sd_vae.VAEConfig):
            sd_vae.build_vae_decoder(sink, cfg)
            return sd_vae.vae_inputs(cfg)
        sd_unet.build_unet(sink, cfg, seed=seed)
        return sd_unet.unet_inputs(cfg, 42)
    g = GraphBuilder(sink, seed=seed)
    ins = case(g)
    # give the graph output (the last op's single output) the stable name "out"
    head, outp = g.lines[-1].split("*output:", 1)
    tok, _, rest = outp.partition("*")
    g.lines[-1] = head + "*output:out" + tok[tok.index("("):] + (("*" + rest) if rest else "")
    g.finish()
    return ins


def all_case_names():
    return [c.__name__ for c in CASES] + [c.__name__ for c in CHAINS] + list(UNETS)


def by_name(name):
    for c in CASES + CHAINS:
        if c.__name__ == name:
            return c
    if name in UNETS:
        return name
    raise KeyError(name)
