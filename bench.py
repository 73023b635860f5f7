#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json): SD 1.5 UNet step latency (ms) + images/sec, 512x512 20-step.

A "step" is ONE denoising step of the reference's txt2img loop (src/sd.cpp:1537-1543): the UNet over the cond and the
uncond sample -- two [1,4,64,64] latents pushed under the same input names (== the reference's m_batch loop,
src/onnxstream.cpp:3847), executed here as one batch-2 pass of the captured hipGraph on inputs that are already resident
in HBM.  W16A16 (fp16 weights, fp16 activations, fp32 accumulation), synthetic graph + seeded random weights of the exact
SD 1.5 topology (no checkpoints offline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N>1: one process per GPU; every rank owns one prompt (its own cond+uncond pass per step) -> weak scaling, no collective on
the data path; RCCL (torch.distributed "nccl") only broadcasts the latents/contexts from rank 0 before the timed region
and gathers the result after it.  Timing: barrier + device sync on both sides of EXACTLY K steps,
MAX over ranks, one JSON line from rank 0.

Extra objects in the JSON line:
  roofline      dominant kernel = the MFMA implicit-GEMM kernel (osg_gemm.hip: Conv + Linear/MatMul/Gemm launches);
                achieved = sum(algorithmic FLOP of its launches in one step) / sum(their HIP-event durations), measured live
                on the backend's compute stream (Model.hip_profile), peak = 2500 TFLOP/s dense fp16 MFMA.
  cpu_baseline  the UNMODIFIED reference (oracle/_ref = reference sources + XNNPACK from libtorch_cpu) timed on this host's
                cores on a bounded sample of the same workload (same model dir, same inputs), rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import re
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

STEPS_PER_IMAGE = 20                 # BASELINE.json: 512x512 20-step
PEAK_F16_TFLOPS = 2500.0             # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
GEMM_KINDS = ("Conv", "Linear", "MatMul", "Gemm", "TBlockTail")   # (TBlockTail: a transformer block's tail as one launch -- six contractions + the 77-token cross-attention, osg_tchain.hip)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cores() -> int:
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def ensure_model_dir(cfg, local_rank, barrier, quant=False):
    from onnxstream_amd.synth import sd_unet
    from onnxstream_amd.synth.graph import DirSink
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), cfg.name + ("_w8" if quant else "")) + "/"
    if local_rank == 0 and not os.path.exists(d + ".complete"):
        t0 = time.time()
        os.makedirs(d, exist_ok=True)
        g, _ = sd_unet.build_unet(DirSink(d), cfg, quant_weights=quant)
        open(d + ".complete", "w").write("ok")
        log(f"[bench] emitted synthetic {cfg.name} UNet: {len(g.lines)} ops, {g.n_params/1e6:.1f} M params in {time.time()-t0:.1f} s")
    barrier()
    return d


def cpu_baseline(model_dir, inputs_cond, inputs_uncond, passes):
    """Reference CPU path on a bounded sample: `passes` UNet passes (cond/uncond alternating) after one cache-filling run,
    with the reference's --ram plumbing (Ram WP + ops cache + next-op cache, src/sd.cpp:1622-1627)."""
    from oracle import ref as oref
    if not oref.available():
        return None
    from onnxstream_amd.bindings import Model
    cores = oref.usable_cores()
    m = Model(oref.REF_LIB, cores, "ram+nocache")
    m.read_file(model_dir + "model.txt")
    m.set_use_ops_cache(True)
    m.set_use_next_op_cache(True)
    times = []
    for i in range(passes + 1):
        ins = inputs_cond if i % 2 == 0 else inputs_uncond
        m.set_use_fp16_arithmetic(False)
        for k, v in ins.items():
            m.add_tensor(k, v)
        m.set_use_fp16_arithmetic(True)
        m.set_fuse_ops_in_attention(True)
        t0 = time.perf_counter()
        m.run()
        times.append(time.perf_counter() - t0)
        m.clear_tensors()
    m.close()
    timed = times[1:]
    pass_s = float(np.mean(timed))
    step_s = 2.0 * pass_s
    return {"value": 1.0 / (STEPS_PER_IMAGE * step_s), "unit": "images/s", "cores": cores, "kind": "reference",
            "ms_per_step": step_s * 1e3,
            "sample": f"{passes} UNet passes (cond/uncond alternating; 2 passes = 1 step) after 1 cache-filling pass "
                      f"({times[0]:.2f} s), W16A16, reference --ram plumbing, images/s = 1/(20 x 2 x {pass_s:.3f} s)"}


PEAK_I8_TOPS = 3944.0                # MI355X dense int8 MFMA (16x16x64), MI355X_MICROARCH.md


def bench_vae_qu8(args):
    """BASELINE config 3, W8A8 half: the fully uint8 SD VAE decoder (exporter layout quant_all, [1,4,64,64] latents -> [1,3,512,512]) with
    m_use_uint8_arithmetic, as `sd --rpi-lowmem` decodes (src/sd.cpp:1212-1222).  range_data.txt comes from a calibration pass of the
    device itself (m_range_data_calibrate; cached beside the synthetic model).  A step = one decode: host quantisation of the pushed
    latents (0.1 % percentiles), upload, the uint8 pass (one launch per graph op), dequantised fp32 image downloaded."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_vae
    from onnxstream_amd.synth.graph import DirSink
    cfg = sd_vae.SD_VAE if args.config == "VAE_QU8" else sd_vae.TINY_VAE
    d = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), cfg.name + "_qu8") + "/"
    if not os.path.exists(d + ".complete"):
        os.makedirs(d, exist_ok=True)
        sd_vae.build_vae_decoder(DirSink(d), cfg, quant_all=True)
        open(d + ".complete", "w").write("ok")
    z = sd_vae.vae_inputs(cfg)[cfg.in_name]
    shipped = os.path.join(REPO, "onnxstream_amd", "synth", "data", cfg.name + "_qu8_range_data.txt")   # a calibration pass of this very (seeded) model, kept
    if os.path.exists(shipped) and not os.path.exists(d + "range_data.txt") and not os.environ.get("OSA_RECALIBRATE"):
        import shutil
        shutil.copy(shipped, d + "range_data.txt")
    if not os.path.exists(d + "range_data.txt"):
        t0 = time.time()
        m = Model(b.LIB_HOST, 0, "ram+nocache")
        m._set_option("range_data_calibrate", 1)
        m.set_use_fp16_arithmetic(True)
        m.read_file(d + "model.txt")
        m.add_tensor(cfg.in_name, z)
        m.run()
        m.hip_write_range_data(d + "range_data.txt")
        m.close()
        log(f"[bench] calibration pass (f16 on the device, percentiles on the host): {time.time()-t0:.1f} s")
        if os.path.isdir(os.path.join(REPO, "gpurun_out")):
            import shutil
            shutil.copy(d + "range_data.txt", os.path.join(REPO, "gpurun_out", cfg.name + "_qu8_range_data.txt"))
    threads = usable_cores()          # the Model's thread count only sets the chunking of the pushed input's percentiles (reference :3091-3104): the reference leg below uses the same
    m = Model(b.LIB_HOST, threads, "ram+nocache")
    m.hip_read_range_data(d + "range_data.txt")
    m.set_use_uint8_arithmetic(True)
    m.read_file(d + "model.txt")

    def decode():
        m.add_tensor(cfg.in_name, z)
        m.run()
        out = m.get_tensor("out_image")[0]
        m.clear_tensors()
        return out
    for _ in range(max(args.warmup, 1)):
        img = decode()
    dev = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = decode()
        dev += m.hip_last_pass_ms()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert np.isfinite(img).all()
    rows = m.hip_profile(args.profile_reps)
    conv = [(ms, fl) for ms, fl, by, what in rows if what.startswith(("Conv qu8", "MatMul qu8 "))]
    c_ms, c_fl = sum(r[0] for r in conv), sum(r[1] for r in conv)
    by_kind = {}
    for ms, fl, by, what in rows:
        k = what.split(" ", 1)[0]
        e = by_kind.setdefault(k, [0.0, 0])
        e[0] += ms; e[1] += 1
    if args.breakdown:
        with open(args.breakdown, "w") as f:
            for k, e in sorted(by_kind.items(), key=lambda kv: -kv[1][0]):
                f.write(f"{k}\t{e[1]}\t{e[0]:.4f}\n")
            for ms, fl, by, what in rows:
                f.write(f"{ms:.5f}\t{fl:.0f}\t{by:.0f}\t{what}\n")
    cpu = None
    from oracle import ref as oref       # (cpu_baseline leg only: the checker, never the thing measured)
    if args.cpu_passes > 0 and oref.available():
        try:
            t0 = time.perf_counter()
            ref_img = oref.run_model_u8(d, {cfg.in_name: z}, open(d + "range_data.txt", newline="").read(), threads=threads)["out_image"]
            s1 = time.perf_counter() - t0
            # the reference's image is there anyway: the timed path must have produced the same codes (the parity gate, inside the bench itself)
            ndiff = int((np.asarray(img) != ref_img).sum())
            if ndiff:
                raise SystemExit(f"bench.py --config {args.config}: the device image differs from the reference's in {ndiff} of {ref_img.size} values (uint8 parity is bit-exact)")
            cpu = {"value": 1.0 / s1, "unit": "decodes/s", "cores": threads, "kind": "reference", "ms_per_step": s1 * 1e3, "output_identical_to_device": True,
                   "sample": "1 decode of the same uint8 model + range data through the reference (m_use_uint8_arithmetic, XNNPACK qu8), model load included; its image equals the device's bit for bit"}
        except Exception as e:
            log(f"[bench] cpu_baseline failed: {e!r}")
    achieved = c_fl / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
    ms_step = wall * 1e3 / args.steps
    line = {"metric": "sd15_vae_decoder_w8a8_decode_latency_ms+decodes_per_sec_512x512", "value": round(1e3 / ms_step, 4), "unit": "decodes/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{cfg.name} decoder, fully uint8 (W8A8, m_use_uint8_arithmetic): [1,4,{cfg.latent},{cfg.latent}] latents -> [1,3,{8*cfg.latent},{8*cfg.latent}], "
                                   "range data from a device calibration pass; per decode: host quantisation of the input, uint8 pass, fp32 image back",
                       "launches_per_step": m.hip_last_kernel_count(), "device_ms_per_step": round(dev / max(args.steps, 1), 4),
                       "by_kind_ms": {k: round(e[0], 4) for k, e in sorted(by_kind.items(), key=lambda kv: -kv[1][0])}},
            "roofline": {"bound": "mfma", "kernel": "q8_gemm_kernel (v_mfma_i32_16x16x64_i8: Conv / MatMul of the uint8 graph)", "achieved": round(achieved, 2),
                         "peak": PEAK_I8_TOPS, "unit": "TOP/s", "frac": round(achieved / PEAK_I8_TOPS, 4), "traffic": None,
                         "launches_per_step": len(conv), "avg_launch_us": c_ms * 1e3 / max(len(conv), 1)},
            "cpu_baseline": cpu}
    m.close()
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="SD15", help="SD15 (headline) | SDXL | TINY | TINY_XL | VAE_QU8 (BASELINE config 3, W8A8 VAE decoder) | VAE_QU8_TINY")
    ap.add_argument("--fusion", type=int, default=2)
    ap.add_argument("--mode", default="pipeline", choices=["pipeline", "replay"],
                    help="pipeline: full txt2img loop (host CFG + Euler-A, VAE decode every 20 steps); replay: UNet graph replays only")
    ap.add_argument("--cpu-passes", type=int, default=4, help="reference CPU passes in the cpu_baseline sample (0 = skip)")
    ap.add_argument("--profile-reps", type=int, default=3)
    ap.add_argument("--steps-per-image", type=int, default=0, help="denoising steps per image (default 20; BASELINE config 4 = SDXL uses 10)")
    ap.add_argument("--no-ln-fold", action="store_true", help="standalone LayerNorm launches instead of folding every LayerNorm into the GEMM that consumes it (osg_gemm_ln, default)")
    ap.add_argument("--no-tblock-fuse", action="store_true", help="the tail of every transformer block as the seven launches of round 3 instead of one osg_tblock_tail launch where it takes the shape (round 4 default)")
    ap.add_argument("--no-concat-views", action="store_true", help="skip tensors through copy launches (Concat) instead of convolutions storing straight into their Concat slot (round 3 default)")
    ap.add_argument("--gn-stats", type=int, default=None, choices=[0, 1, 2], help="GroupNorm statistics from the producing convolutions' epilogues: 0 never, 1 every eligible GroupNorm, 2 only tensors of >= 8 M elements (the Model's default: pays in the throughput regime, neutral on the SD 1.5 pass, profiles/r03_gn_stats_ab.txt)")
    ap.add_argument("--frozen-table", action="store_true", help="N = 1 as the ranks of an N > 1 job run: the shipped tune table, OSG_TUNE_FROZEN=1 (a shape it does not hold takes the cost model's first candidate, nothing is timed) -- the same plan at every N")
    ap.add_argument("--no-autotune", action="store_true", help="tile / split-K configurations from the cost model only (no measured choice in the first pass)")
    ap.add_argument("--host-loop", action="store_true", help="pipeline mode: CFG + Euler-A on the host with one round trip per step (the reference app's shape) instead of the device loop")
    ap.add_argument("--prompts-per-gpu", type=int, default=1, help="prompts denoised together on each GPU (2P samples per UNet pass: the reference's --num batching)")
    ap.add_argument("--w8-resident", action="store_true", help="with --quant-weights: keep the codes resident and dequantise on chip (osg_*_w8 kernels)")
    ap.add_argument("--quant-weights", action="store_true", help="W8A16: uint8 weights + scale/zero-point in model.txt, dequantised at load")
    ap.add_argument("--clamp", action="store_true", help="clamp the latent to +-4 max(sigma, 1) after every step (NOT in the reference loop; rounds 1 used it to keep "
                    "random-weight trajectories finite -- they stay finite without it, see config.latent_absmax)")
    ap.add_argument("--breakdown", default="", help="write the per-step HIP-event profile to this file")
    ap.add_argument("--windows", type=int, default=5, help="further one-image windows timed after the K steps (spread only: median / min / max in config.windows_ms_per_step; 0 = none)")
    args = ap.parse_args()
    if args.config in ("VAE_QU8", "VAE_QU8_TINY"):
        return bench_vae_qu8(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with `python -m torch.distributed.run --nproc-per-node N`")
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP backend has no CPU fallback")
    # OSA_BENCH_ONE_GPU=1 (dev check of the N > 1 flow on a 1-GPU box, tools/gpu_round.sh tworank): every rank drives cuda:0 and the process group is
    # gloo over host tensors (RCCL refuses two ranks on one device).  Never set by the driver; the line it prints says so ("dev_one_gpu").
    one_gpu = os.environ.get("OSA_BENCH_ONE_GPU") == "1"
    gpu_index = 0 if one_gpu else local_rank
    coll_dev = "cpu" if one_gpu else "cuda"
    torch.cuda.set_device(gpu_index)
    dist = None
    # OSA_BENCH_FORCE_DIST=1 (dev check on a 1-GPU box): a world of ONE rank still goes through the process group, so every RCCL call of the N > 1 flow
    # (init with device_id, broadcast / all_gather / all_reduce / barrier on device tensors) is exercised on the real backend
    if world > 1 or os.environ.get("OSA_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    from onnxstream_amd import build as b
    from onnxstream_amd.bindings import Model
    from onnxstream_amd.synth import sd_unet
    if not (os.path.exists(b.LIB_GPU) and os.path.exists(b.LIB_HOST)):
        raise SystemExit("native libraries missing: run `python __graft_entry__.py` (build()) first")

    cfg = getattr(sd_unet, args.config)
    model_dir = ensure_model_dir(cfg, local_rank, barrier, args.quant_weights)
    global STEPS_PER_IMAGE
    if args.steps_per_image > 0:
        STEPS_PER_IMAGE = args.steps_per_image

    # ---- inputs: rank 0 draws every prompt's latents/contexts, RCCL-broadcasts them over xGMI, each rank keeps its own ------
    # (one prompt per rank; a prompt = its cond and its uncond sample, stacked on a leading axis of 2)
    from onnxstream_amd import shard

    def draw(i):
        c, u = sd_unet.unet_inputs(cfg, 42 + 2 * i), sd_unet.unet_inputs(cfg, 43 + 2 * i)
        return {k: np.stack([c[k], u[k]]) for k in c}
    mine = shard.scatter_prompts(dist, rank, world, world, draw, device=coll_dev if dist is not None else "cpu")
    (my_prompt, pack), = mine.items()
    cond = {k: v[0] for k, v in pack.items()}
    uncond = {k: v[1] for k, v in pack.items()}

    # ---- the product path: model_* C API -> host planner -> libosgpu HIP kernels, driven by the txt2img harness ---------------
    from onnxstream_amd.pipeline import Txt2Img, sigma_schedule
    from onnxstream_amd.synth import sd_vae
    import dataclasses
    vcfg = dataclasses.replace(sd_vae.SD_VAE, latent=cfg.latent, name=f"sd_vae{cfg.latent}") if cfg.latent >= 64 else \
        dataclasses.replace(sd_vae.TINY_VAE, latent=cfg.latent, name=f"tiny_vae{cfg.latent}")
    vae_dir = None
    if args.mode == "pipeline":
        from onnxstream_amd.synth.graph import DirSink
        vae_dir = os.path.join(os.environ.get("OSA_SYNTH_DIR", "/tmp/onnxstream_amd_synth"), vcfg.name) + "/"
        if local_rank == 0 and not os.path.exists(vae_dir + ".complete"):
            sd_vae.build_vae_decoder(DirSink(vae_dir), vcfg)
            open(vae_dir + ".complete", "w").write("ok")
        barrier()
    t_build = time.time()
    # measured tile / ring / split-K choices: seeded from the table shipped with the repo (onnxstream_amd/tune/mi355x.txt: measured on an MI355X with the
    # kernels of that commit; rows the library no longer offers are dropped on load, shapes it does not cover are measured now and appended to the
    # process-private copy) unless the caller names its own OSG_TUNE_CACHE -- every rank of a node then starts from the same choices
    shipped_table = os.path.join(REPO, "onnxstream_amd", "tune", "mi355x.txt")
    tune_src = "none (cost model)" if args.no_autotune else ("OSG_TUNE_CACHE of the caller" if os.environ.get("OSG_TUNE_CACHE") else
                                                             ("shipped table onnxstream_amd/tune/mi355x.txt, missing shapes measured now (cold operands)" if os.path.exists(shipped_table)
                                                              else "measured in the plan-building pass (cold operands)"))
    if not args.no_autotune and not os.environ.get("OSG_TUNE_CACHE") and dist is None and not args.frozen_table:
        import shutil
        private = f"/tmp/osg_tune_{os.getuid()}_{os.getpid()}.txt"
        if os.path.exists(shipped_table):
            shutil.copy(shipped_table, private)
        os.environ["OSG_TUNE_CACHE"] = private

    def make_pipe():
        return Txt2Img(b.LIB_HOST, model_dir, vae_dir, batched=True, device=gpu_index, fusion=args.fusion, autotune=not args.no_autotune)
    pipe = None
    frozen_ranks = (dist is not None or args.frozen_table) and not args.no_autotune and (args.frozen_table or not os.environ.get("OSG_TUNE_CACHE")) and os.path.exists(shipped_table)
    if frozen_ranks:
        # N > 1, no table named by the caller: every rank seeds a private copy of the SHIPPED table and plans on its own with OSG_TUNE_FROZEN = 1 -- a shape
        # the table does not hold takes the cost model's first candidate (deterministic) instead of being timed, so all ranks make identical choices
        # without rank 0 measuring anything while the others wait (round-3 review: share_tune_table serialised the start-up on rank 0)
        import shutil
        private = f"/tmp/osg_tune_{os.getuid()}_{os.getpid()}_rank{rank}.txt"
        shutil.copy(shipped_table, private)
        os.environ["OSG_TUNE_CACHE"] = private
        os.environ["OSG_TUNE_FROZEN"] = "1"
        tune_src = "shipped table onnxstream_amd/tune/mi355x.txt on every rank, frozen (missing shapes: deterministic cost-model choice, nothing timed)"
    elif dist is not None and not args.no_autotune:
        # N > 1: every rank must make the SAME measured tile / split-K choices (else the same prompt gives different last bits on different
        # GPUs): rank 0 plans + tunes first, its table travels over RCCL, the other ranks start seeded from it and time nothing
        def tune_on_rank0():
            nonlocal pipe
            pipe = make_pipe()
            x0 = np.random.default_rng(7).standard_normal((1, cfg.in_ch, cfg.latent, cfg.latent), dtype=np.float32)
            pipe.denoise(x0, 1.0, cond["encoder_hidden_states"], uncond["encoder_hidden_states"],
                         extra_cond={k: cond[k] for k in ("text_embeds", "time_ids") if k in cond} or None,
                         extra_uncond={k: uncond[k] for k in ("text_embeds", "time_ids") if k in uncond} or None)
            if pipe.vae is not None:
                pipe.decode(x0)
        shard.share_tune_table(dist, rank, world, f"/tmp/osg_tune_shared_rank{rank}_{os.getpid()}.txt", tune_on_rank0, device=coll_dev,
                               seed=None if os.environ.get("OSG_TUNE_CACHE") else shipped_table)
    if pipe is None:
        pipe = make_pipe()
    m = pipe.unet
    if args.w8_resident:
        m._set_option("hip_w8_resident", 1)
    if args.no_ln_fold:
        m._set_option("hip_fuse_ln_gemm", 0)
    if args.no_concat_views:
        m._set_option("hip_concat_views", 0)
    if args.no_tblock_fuse:
        m._set_option("hip_fuse_tblock", 0)
    if args.gn_stats is not None:
        m._set_option("hip_gn_stats", args.gn_stats)
    L = cfg.latent
    P = max(1, args.prompts_per_gpu if args.mode == "pipeline" and not cfg.sdxl_add_embed else 1)
    lat_shape = (P, cfg.in_ch, L, L)
    ctx_c, ctx_u = cond["encoder_hidden_states"], uncond["encoder_hidden_states"]
    # P prompts per GPU share the schedule; prompt p's contexts are a fixed perturbation of this rank's (synthetic data)
    ctx_cs = ctx_c if P == 1 else [np.roll(ctx_c, p_i, axis=1) for p_i in range(P)]
    ctx_us = ctx_u if P == 1 else [np.roll(ctx_u, p_i, axis=1) for p_i in range(P)]
    ex_c = {k: cond[k] for k in ("text_embeds", "time_ids") if k in cond} or None
    ex_u = {k: uncond[k] for k in ("text_embeds", "time_ids") if k in uncond} or None
    sig = sigma_schedule(STEPS_PER_IMAGE, pipe.log_sigmas)
    rng = np.random.default_rng(1234 + rank)
    state = {"x": rng.standard_normal(lat_shape, dtype=np.float32) * sig[0], "i": 0, "images": 0, "last": None}

    host_loop = args.host_loop
    trace = [] if os.environ.get("OSA_BENCH_TRACE") else None     # dev: (what, steps, host ms in front of the call, wall ms of the call, device ms) per chunk -> stderr
    from concurrent.futures import ThreadPoolExecutor
    noise_pool, noise_rng = ThreadPoolExecutor(1), np.random.default_rng(99 + rank)
    scal = pipe.loop_scalars(sig)
    # the timed loop is the reference's arithmetic and nothing else: no clamp unless --clamp (a random-weight UNet does not denoise, its
    # latents stay at the initial noise scale ~ sigma_0 instead of shrinking to ~1; they stay finite, asserted after the timed region)
    clip = np.asarray([4.0 * max(float(sig[i + 1]), 1.0) for i in range(STEPS_PER_IMAGE)], np.float32) if args.clamp else None
    n_names = pipe.names

    def end_of_image(x):
        if pipe.vae is not None:
            state["last"] = pipe.decode(x)
        state["images"] += P
        return rng.standard_normal(lat_shape, dtype=np.float32) * sig[0]

    def run_steps(k):
        """k denoising steps of the 20-step loop: per step the UNet over cond+uncond as one batch-2P pass, the CFG combine and the
        Euler-Ancestral update; after the 20th step of an image the VAE decode, then a fresh latent.  Default: the loop runs on the
        device (Model.hip_sampler_loop: scaling kernel -> captured pass -> CFG/Euler-A kernel per step, one host sync per image);
        --host-loop: sampler arithmetic on the host exactly as the reference app does (one upload/sync/download per step).
        Returns the device ms spent in the steps."""
        dev = 0.0
        while k > 0:
            i = state["i"]
            tr_in = time.perf_counter()
            if args.mode == "replay":
                m.hip_replay(1)
                dev += m.hip_last_pass_ms()
                state["i"] = (i + 1) % STEPS_PER_IMAGE
                k -= 1
                continue
            if host_loop:
                x = state["x"]
                den = pipe.denoise(x, float(sig[i]), ctx_cs, ctx_us, extra_cond=ex_c, extra_uncond=ex_u)
                dev += m.hip_last_pass_ms()
                x = (x + ((x - den) / scal[3][i]) * scal[4][i] + rng.standard_normal(lat_shape, dtype=np.float32) * scal[5][i]).astype(np.float32)
                if clip is not None:
                    x = np.clip(x, -clip[i], clip[i])
                n = 1
            else:
                n = min(k, STEPS_PER_IMAGE - i)
                x = np.ascontiguousarray(state["x"], np.float32)
                # the ancestral noise of this chunk was drawn on a worker thread while the previous chunk ran on the GPU
                fut = state.pop("noise", None)
                noise = fut.result() if fut is not None and fut.n == n else noise_rng.standard_normal((n,) + lat_shape, dtype=np.float32)
                rem_img = STEPS_PER_IMAGE - (i + n) % STEPS_PER_IMAGE
                nxt = min(k - n, rem_img) if k > n else rem_img
                state["noise"] = noise_pool.submit(noise_rng.standard_normal, (nxt,) + lat_shape, np.float32)
                state["noise"].n = nxt
                tr0 = time.perf_counter()
                d1 = m.hip_sampler_loop(n_names["sample"], n_names["timestep"], n_names["out"], x, noise, *[a[i:i + n] for a in scal], 7.0, clip[i:i + n] if clip is not None else None)
                dev += d1
                if trace is not None:
                    trace.append(("loop", n, (tr0 - tr_in) * 1e3, (time.perf_counter() - tr0) * 1e3, d1))
            if i + n == STEPS_PER_IMAGE:
                tr0 = time.perf_counter()
                x = end_of_image(x)
                if trace is not None:
                    trace.append(("end_of_image", 0, 0.0, (time.perf_counter() - tr0) * 1e3, pipe.vae.hip_last_pass_ms() if pipe.vae is not None else 0.0))
            state["x"], state["i"] = x, (i + n) % STEPS_PER_IMAGE
            k -= n
        return dev

    # plan + eager pass (weights become resident), hipGraph capture, and the VAE's plan/capture: all before the timed region
    if args.mode == "replay":
        for r in range(2):
            for ins in (cond, uncond):
                for k, v in ins.items():
                    m.add_tensor(k, v)
            m.set_use_fp16_arithmetic(True)
            m.set_fuse_ops_in_attention(True)
            m.run()
            if r == 0:
                m.clear_tensors()
    else:
        for _ in range(3):
            pipe.denoise(state["x"], float(sig[0]), ctx_cs, ctx_us, extra_cond=ex_c, extra_uncond=ex_u)
        if pipe.vae is not None:
            for _ in range(3):
                pipe.decode(state["x"] / sig[0])
    kernels = m.hip_last_kernel_count()
    vae_kernels = pipe.vae.hip_last_kernel_count() if pipe.vae is not None else 0
    log(f"[bench r{rank}] plan+capture {time.time()-t_build:.1f} s, UNet {kernels} launches/pass ({m.hip_last_pass_ms():.3f} ms device)"
        + (f", VAE {vae_kernels} launches ({pipe.vae.hip_last_pass_ms():.3f} ms device)" if pipe.vae is not None else ""))

    # ---- warmup, then EXACTLY K timed steps ---------------------------------------------------------------------------
    run_steps(args.warmup)
    state["i"], state["images"] = 0, 0           # the timed region starts at the first step of an image
    if not host_loop and args.mode != "replay":
        # ... with its ancestral noise drawn, as every later chunk finds it (drawn on the worker thread while the chunk before runs): the future the warmup
        # left behind is for the REST of the warmup's image, its size does not match a chunk starting at step 0 and the main thread would draw inside the timed region
        n0 = min(args.steps, STEPS_PER_IMAGE)
        state["noise"] = noise_pool.submit(noise_rng.standard_normal, (n0,) + lat_shape, np.float32)
        state["noise"].n = n0
        state["noise"].result()
    if trace is not None:
        trace.clear()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    dev_acc = run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = dev_acc / max(args.steps, 1)
    # ---- spread: `--windows` further windows of one image each (STEPS_PER_IMAGE steps + decode), timed the same way AFTER the contract's K steps.  They do
    # not enter `value` / `ms_per_step`; they show what a 2 % change looks like against the run-to-run spread of this box.
    images_timed = state["images"]
    win_ms = []
    for _ in range(max(0, args.windows)):
        state["i"] = 0
        torch.cuda.synchronize()
        tw0 = time.perf_counter()
        run_steps(STEPS_PER_IMAGE)
        torch.cuda.synchronize()
        win_ms.append((time.perf_counter() - tw0) * 1e3 / STEPS_PER_IMAGE)
    if trace is not None:
        for tr in trace:
            log("[bench trace] %-13s steps %2d  host before %.3f ms  call %.3f ms  device %.3f ms" % tr)
    out = state["last"] if state["last"] is not None else state["x"]
    latent_absmax = float(np.abs(state["x"]).max())
    if not (np.isfinite(np.asarray(out, np.float32)).all() and np.isfinite(latent_absmax)):
        raise SystemExit("bench.py: non-finite latents / image after the timed region")
    per_rank_ms = None
    if dist is not None:
        # every rank's own time of the K steps (a straggler GPU is visible in the line), then the MAX the contract asks for
        mine_t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        per_rank_ms = [round(float(t.item()) * 1e3 / args.steps, 4) for t in allt]
        tw = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        # gather every prompt's result (image, or latents when no image completed) on rank 0
        allr = shard.gather_results(dist, rank, world, world, {my_prompt: np.asarray(out, np.float32)}, device=coll_dev)
        assert rank != 0 or (allr.shape[0] == world and np.isfinite(allr).all())
    ms_per_step = wall * 1e3 / args.steps
    images_per_s = world * P / (STEPS_PER_IMAGE * ms_per_step * 1e-3)
    # shapes the measured-choice table did not hold (timed live at N = 1, cost-model choice under OSG_TUNE_FROZEN): 0 means N = 1 and N > 1 time the same plan
    try:
        import ctypes
        tune_misses = int(ctypes.CDLL(b.LIB_GPU).osg_tune_misses()) if not args.no_autotune else None
    except Exception:
        tune_misses = None

    line = None
    if rank == 0:
        # ---- roofline of the dominant kernel, live (HIP events on the backend's compute stream) ---------------------------
        rows = m.hip_profile(args.profile_reps)
        by_kind = {}
        for ms, fl, by, what in rows:
            k = what.split(" ", 1)[0].split("+", 1)[0]
            e = by_kind.setdefault(k, [0.0, 0.0, 0.0, 0])
            e[0] += ms; e[1] += fl; e[2] += by; e[3] += 1
        g_ms = sum(by_kind[k][0] for k in GEMM_KINDS if k in by_kind)
        g_fl = sum(by_kind[k][1] for k in GEMM_KINDS if k in by_kind)
        g_n = sum(by_kind[k][3] for k in GEMM_KINDS if k in by_kind)
        tot_fl = sum(r[1] for r in rows)
        tot_ms = sum(r[0] for r in rows)
        # Per-launch HIP events exist only for an EAGER pass, and eager launches are ~0.3-0.6 us longer each than the same nodes of the captured pass the contract
        # times: the eager sum exceeded the driver-timed step in round 5.  The contraction kernels' time is therefore their SHARE of the eager
        # pass applied to the device time of the CAPTURED pass (dev_ms, HIP events around the replayed graph): live, and never more than the step it is part of.
        g_ms_eager = g_ms
        if tot_ms > 0 and dev_ms > 0:
            g_ms = dev_ms * (g_ms_eager / tot_ms)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        # Counter-derived figures of the contraction kernels from THIS round's rocprofv3 --pmc passes (tools/pmc_round.sh over eager passes of
        # the same plan; FETCH_SIZE and WRITE_SIZE collected in separate passes, KiB; FETCH_SIZE doubled per the gfx950 correction of the
        # microarch guide; SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines x 256 CUs x 4 SIMDs) = share of the SIMD-cycles of the
        # kernels' own run time in which the matrix pipe was busy).  null when no counter file of the round is committed.
        traffic = mfma_util = hbm_gbs = None
        pmc_src = pmc_note = None
        # only a counter file of THIS round whose own header says it was collected on the tuned plan (tools/pmc_round3.sh: same OSG_TUNE_CACHE table as the
        # timed run, eager passes of that plan) describes the kernels that are timed here; round 2's file was taken with autotune off and is NOT used any
        # more.  No such file => the three counter fields are null.
        # round 4: ... AND whose `src_sha1` is the hash of the kernel / host sources of THIS tree (tools/src_hash.py; tools/pmc_round4.sh records it): counters of
        # another tree's kernels are not reported as this run's (the round-3 file described a mid-round commit)
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import src_hash
        tree = src_hash.src_sha1(REPO)
        for cand in sorted((f for f in os.listdir(os.path.join(REPO, "profiles")) if re.match(r"r\d\d_pmc.*\.json$", f)), reverse=True):   # (newest round first; the src_sha1 test below picks the file of THIS tree)
            try:
                hdr = json.load(open(os.path.join(REPO, "profiles", cand)))
            except Exception:
                continue
            if hdr.get("tuned_plan") and "kernels" in hdr and hdr.get("src_sha1") == tree and bool(hdr.get("w8_resident")) == bool(args.w8_resident):   # (the W8-resident plan has a counter file of its own)
                pmc_src, pmc_note = cand, hdr.get("note")
                break
        if pmc_src and cfg.name == "sd15" and P == 1 and not args.no_autotune:   # (the counter file describes the tuned batch-2 pass of one prompt)
            tj = json.load(open(os.path.join(REPO, "profiles", pmc_src)))["kernels"]
            sel = [v for k, v in tj.items() if "gemm2_kernel" in k or "conv3x3_kernel" in k or "conv_small" in k or "gemm_kernel" in k or "splitk_reduce" in k or "tblock_tail" in k or "conv_cin4" in k]
            nd = sum(v["dispatches"] for k, v in tj.items() if ("gemm2_kernel" in k or "conv3x3_kernel" in k or "conv_small" in k or "gemm_kernel" in k or "tblock_tail" in k or "conv_cin4" in k))
            if nd:
                traffic = (sum(2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0) for v in sel) * 1024.0) / nd
                busy = sum(v.get("SQ_BUSY_CYCLES", 0.0) for v in sel)
                if busy > 0:
                    mfma_util = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in sel) / (busy / 32.0 * 1024.0)
                if g_ms > 0 and g_n:
                    hbm_gbs = traffic / (g_ms * 1e-3 / g_n) / 1e9
        roofline = {"bound": "mfma", "kernel": "gemm2_kernel + conv3x3_kernel + tblock_tail_kernel (implicit-GEMM / halo-reuse Conv, Linear/MatMul/Gemm, fused transformer-block tails)", "achieved": round(achieved, 2),
                    "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": f"HBM-side bytes per contraction launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of profiles/{pmc_src}, split-K reduce launches folded in; {pmc_note})" if traffic is not None else None,
                    "counters": (f"profiles/{pmc_src}: eager passes of the tuned plan (same tune table as the timed hipGraph run), one counter set per rocprofv3 pass" if traffic is not None
                                 else "null: no counter file collected on the timed plan of THIS tree is committed (tools/pmc_round4.sh makes one; its src_sha1 must match tools/src_hash.py)"),
                    "mfma_util": round(mfma_util, 4) if mfma_util is not None else None,
                    "hbm_gbs": round(hbm_gbs, 1) if hbm_gbs is not None else None, "hbm_frac": round(hbm_gbs / PEAK_HBM_GBS, 4) if hbm_gbs is not None else None,
                    "launches_per_step": g_n, "flop_per_launch": g_fl / max(g_n, 1), "avg_launch_us": g_ms * 1e3 / max(g_n, 1),
                    "step_flop": tot_fl, "step_frac": round(tot_fl / (ms_per_step * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                    "sum_of_kernels_ms": round(tot_ms, 4),
                    "method": ("contraction time = (their share of an eager pass timed launch by launch with HIP events on the backend's stream) x (device time of the captured, replayed pass); "
                               "the in-graph rocprofv3 timeline of the same tree: profiles/ (tools/graph_trace.py)"),
                    "contraction_ms_eager_events": round(g_ms_eager, 4), "contraction_ms_in_captured_pass": round(g_ms, 4), "captured_pass_ms": round(dev_ms, 4)}
        if args.breakdown:
            with open(args.breakdown, "w") as f:
                f.write("# kind\tlaunches\tms\tGFLOP\tTFLOP/s\tGB(read+write of operands)\tGB/s\n")
                for k, e in sorted(by_kind.items(), key=lambda kv: -kv[1][0]):
                    f.write(f"{k}\t{e[3]}\t{e[0]:.4f}\t{e[1]/1e9:.2f}\t{(e[1]/(e[0]*1e-3)/1e12 if e[0] else 0):.1f}\t{e[2]/1e9:.4f}\t{(e[2]/(e[0]*1e-3)/1e9 if e[0] else 0):.1f}\n")
                f.write("# per step: ms\tflops\tbytes\twhat\n")
                for ms, fl, by, what in rows:
                    f.write(f"{ms:.5f}\t{fl:.0f}\t{by:.0f}\t{what}\n")
        cpu = None
        if world == 1 and args.cpu_passes > 0:
            pipe.close()
            pipe = None
            try:
                cpu = cpu_baseline(model_dir, cond, uncond, args.cpu_passes)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                log(f"[bench] cpu_baseline failed: {e!r}")
        line = {
            "metric": "sd15_unet_step_latency_ms+images_per_sec_512x512_20step" if cfg.name == "sd15" else f"{cfg.name}_unet_step_latency_ms+images_per_sec_{STEPS_PER_IMAGE}step", "value": round(images_per_s, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # the headline-of-record against box-to-box / window-to-window spread: the median of the further one-image windows (config.windows_ms_per_step)
            "ms_per_step_window_median": round(float(np.median(win_ms)), 4) if win_ms else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": (f"{cfg.name} {8 * cfg.latent}x{8 * cfg.latent} {STEPS_PER_IMAGE}-step txt2img ({'W8A16' if args.quant_weights else 'W16A16'}): per step the UNet over cond+uncond (2x4x{cfg.latent}x{cfg.latent} latents, ctx 77x{cfg.ctx_dim}) "
                                    f"as one batch-2 pass + CFG 7 + Euler-Ancestral update ({'on the host, one round trip per step' if args.host_loop else 'on the device, one host sync per image'}), VAE decode after the last step of every image (inside the timed region), "
                                    f"weights resident; ms_per_step = wall / K with the decode amortised; images/s = gpus x prompts_per_gpu / (steps_per_image x ms_per_step)")
                                   if vae_dir else (f"{cfg.name} UNet denoising step: cond+uncond 2x4x{cfg.latent}x{cfg.latent} latents, W16A16, "
                                                    f"weights resident, mode={args.mode}; NO VAE decode"),
                       "mode": args.mode, "sampler": "host" if args.host_loop else "device", "autotune": not args.no_autotune, "tune_table": tune_src, "tune_table_frozen": bool(frozen_ranks), "tune_table_misses": tune_misses, "vae_decode_in_timed_region": bool(vae_dir), "images_completed": images_timed, "clamp": bool(args.clamp), "latent_absmax": round(latent_absmax, 3),
                       "prompts_per_gpu": P, "unet_passes_per_step": 2 * P, "steps_per_image": STEPS_PER_IMAGE, "launches_per_step": kernels,
                       "vae_launches": vae_kernels, "fusion_level": args.fusion, "unet_device_ms_per_step": round(dev_ms, 4),
                       "parallelism": f"replica x{world}" + (" (dev_one_gpu: all ranks on cuda:0, gloo)" if one_gpu else ""),
                       "per_rank_ms_per_step": per_rank_ms,
                       "windows_ms_per_step": ({"n": len(win_ms), "each": [round(w, 4) for w in win_ms], "median": round(float(np.median(win_ms)), 4), "min": round(min(win_ms), 4),
                                               "max": round(max(win_ms), 4), "what": f"{len(win_ms)} further one-image windows ({STEPS_PER_IMAGE} steps + decode) on rank 0 after the K timed steps; not part of value"}
                                               if win_ms else None)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    if pipe is not None:
        pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
