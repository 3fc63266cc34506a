#!/usr/bin/env python
"""Headline benchmark: img/s of forward + backward of the full OneFormer model with a Swin-L backbone
on synthetic 1024x2048 batches, 2 images per GPU, data-parallel over N GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = zero grads, re-cast the fp32 master weights to bf16 operands (as after an optimizer step),
forward of backbone + pixel decoder + transformer decoder + the final x4 mask upsample
(reference model/oneformer_model.py:244-263), the synthetic deep-supervision loss of SURVEY.md §8d,
backward, and (N > 1) the bucketed RCCL gradient all-reduce.  Rank 0 prints ONE JSON line.
The timed region is bracketed by barrier + torch.cuda.synchronize() on both sides; the reported time
is the MAX over ranks; `value` = images all ranks processed / that time.  The headline runs the model in eval() mode
(deterministic, the full FLOPs of every residual branch); the training-mode step (DropPath / dropout on) is timed as
well and reported under `train_mode`.

Extra keys of the line (all measured in this run; rank 0, N = 1 only, after the timed region):
  `step_ms`            per-iteration device time of the timed steps (HIP events): median / min / max
  `host_enqueue_ms`    host time to enqueue one step from an idle stream (launch-bound if >= ms_per_step)
  `forward_only`       inference forward (no_grad, incl. the mask upsample) img/s
  `train_mode`         the same step with model.train(): stochastic depth + the decoders' dropout active
  `gpu_eager_baseline` oracle/torch_ref.py fp32 forward on the SAME GPU through ATen ("reference PyTorch-ROCm path" of
                       BASELINE.json's north_star; a baseline, not the product) and the product's forward speed-up over it
  `roofline`           dominant kernel family (bf16 MFMA GEMM) timed per launch with HIP events on its launch stream
                       (`uenc_prof_*`) over a repetition of the timed steps; algorithmic FLOPs = 2*M*N*K per launch;
                       per-family algorithmic bytes; `traffic` from the committed rocprofv3 PMC summary when it was
                       measured on the current kernel sources (else null)
  `cpu_baseline`       the fp32 oracle (a "port") on the host cores: forward+backward of ONE full-size 1024x2048 image, two samples;
                       `c0` = BASELINE configs[0] (Swin-T, 1 x 512 x 1024, forward, 1 warm-up + 3 timed)
"""
import argparse
import ctypes
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "uni-encoder-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist
import torch.nn.functional as F

H_IMG, W_IMG, PER_GPU_BATCH = 1024, 2048, 2
SWIN_L = dict(EMBED_DIM=192, DEPTHS=[2, 2, 18, 2], NUM_HEADS=[6, 12, 24, 48], WINDOW_SIZE=12)
# forward FLOPs per image of the workload: backbone 3109.3 GF + head 866.6 GF; fwd+bwd = 3x
GFLOP_FWD_PER_IMG = 3976.0
PROFILE_ROUND = "r03"


# BASELINE configs[4] (not the headline): UENC_BENCH_BACKBONE=dinat swaps the backbone for DiNAT-L (kernel 7; dilations = 1 / the largest
# that fits each stage of a 1024 x 2048 input) and tags the JSON line; metric, loss and protocol stay the same.
DINAT_L = dict(EMBED_DIM=192, MLP_RATIO=2.0, DEPTHS=[3, 4, 18, 5], NUM_HEADS=[6, 12, 24, 48], KERNEL_SIZE=7,
               DILATIONS=[[1, 16, 1], [1, 8, 1, 8], [1, 4] * 9, [1, 2, 1, 2, 1]])
BACKBONE = os.environ.get("UENC_BENCH_BACKBONE", "swin").lower()


def make_cfg(device):
    import model  # noqa: F401  (registers OneFormer, D2SwinTransformer, OneFormerHead, decoders)
    from uenc.config import add_common_config, add_dinat_config, add_swin_config, add_uni_encoder_config
    from uenc.d2 import get_cfg
    cfg = get_cfg()
    add_common_config(cfg); add_swin_config(cfg); add_dinat_config(cfg); add_uni_encoder_config(cfg)
    opts = ["MODEL.META_ARCHITECTURE", "OneFormer", "MODEL.BACKBONE.NAME", "D2SwinTransformer",
            "MODEL.SEM_SEG_HEAD.NAME", "OneFormerHead", "MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME", "MSDeformAttnPixelDecoder",
            "MODEL.SEM_SEG_HEAD.NUM_CLASSES", 19, "MODEL.SEM_SEG_HEAD.CONVS_DIM", 256, "MODEL.SEM_SEG_HEAD.MASK_DIM", 256,
            "MODEL.SEM_SEG_HEAD.IN_FEATURES", ["res2", "res3", "res4", "res5"], "MODEL.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS", 6,
            "MODEL.ONE_FORMER.TRANSFORMER_IN_FEATURE", "multi_scale_pixel_decoder", "MODEL.ONE_FORMER.NUM_OBJECT_QUERIES", 150,
            "MODEL.ONE_FORMER.DEC_LAYERS", 10, "MODEL.IS_TRAIN", False,
            "MODEL.PIXEL_MEAN", [123.675, 116.280, 103.530], "MODEL.PIXEL_STD", [58.395, 57.120, 57.375],
            "MODEL.DEVICE", device]
    for k, v in SWIN_L.items():
        opts += [f"MODEL.SWIN.{k}", v]
    if BACKBONE == "dinat":
        opts[opts.index("D2SwinTransformer")] = "D2DiNAT"
        for k, v in DINAT_L.items():
            opts += [f"MODEL.DiNAT.{k}", v]
    cfg.merge_from_list(opts)
    return cfg


class _WeightedMeanSquares(torch.autograd.Function):
    """sum_i w_i * mean(x_i^2) over a list of tensors as ONE autograd node: the norms of all tensors by multi-tensor kernels
    (torch._foreach_norm), the gradient of each tensor as one scaled copy.  (Per-tensor pow / mean / mul / add chains cost ~130
    tiny launches per step for the 20 predictions, and autograd's pow-mean chain five passes over every 157 MB mask tensor.)"""

    @staticmethod
    def forward(ctx, weights, *xs):
        ctx.save_for_backward(*xs)
        norms = torch.stack(torch._foreach_norm([x.detach().float() if x.dtype != torch.float32 else x.detach() for x in xs]))
        coef = torch.tensor([w / x.numel() for w, x in zip(weights, xs)], dtype=torch.float32).to(norms.device, non_blocking=True)
        ctx.coef = coef
        return (norms.square() * coef).sum()

    @staticmethod
    def backward(ctx, g):
        xs = ctx.saved_tensors
        c = (ctx.coef * (2.0 * g)).unbind()          # one launch for the 20 coefficients
        # multi-tensor scaled copies (one launch per dtype group); `x * c_i` with a 0-dim device tensor runs the un-vectorised broadcast kernel
        out = [None] * len(xs)
        for dt in {x.dtype for x in xs}:
            idx = [i for i, x in enumerate(xs) if x.dtype == dt]
            for i, y in zip(idx, torch._foreach_mul([xs[i] for i in idx], [c[i].to(dt) for i in idx])):
                out[i] = y
        return (None,) + tuple(out)


def synthetic_loss(out):
    """mean-square of logits and masks, x0.1 on the nine auxiliary predictions."""
    xs, ws = [out["pred_logits"], out["pred_masks"]], [1.0, 1.0]
    for a in out["aux_outputs"]:
        xs += [a["pred_logits"], a["pred_masks"]]
        ws += [0.1, 0.1]
    return _WeightedMeanSquares.apply(ws, *xs)


def _cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(cores, 16)        # a 1-GPU box grants 16 CPUs whatever the host has: more threads only oversubscribe them


def cpu_baseline():
    """The fp32 oracle (a "port" of the reference's path, validated against the reference's modules on the committed fixtures) on
    the host cores, per BASELINE.md §4 / SURVEY.md §8(d):
      * headline sample (same unit as `value`): forward + backward of the Swin-L model on ONE full-size 1024 x 2048 image, timed
        TWICE after a warm-up (both samples reported; a bs-2 iteration would be 2 x ~40 s per sample, so one image is the unit and
        nothing is extrapolated);
      * `c0`: BASELINE configs[0] -- full model with the Swin-T backbone, one 512 x 1024 image, forward only, 1 warm-up + 3 timed.
    About 100 s of CPU work in total."""
    from oracle import fill, torch_ref as T
    cores = _cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)

    # -- C0: Swin-T, 1 x 512 x 1024, forward
    cfg0 = T.ModelCfg(swin=T.SwinCfg(96, (2, 2, 6, 2), (3, 6, 12, 24), 7))
    sd0 = fill.state_dict_for(T.model_param_shapes(cfg0))
    img0 = torch.randint(0, 256, (3, 512, 1024), generator=g).float()
    print(f"[bench] cpu_baseline: oracle on {cores} threads, configs[0] Swin-T 512x1024 forward x (1 + 3) ...", file=sys.stderr, flush=True)
    t_c0 = []
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            T.oneformer_forward([{"left_image": img0, "task": "The task is panoptic"}], sd0, cfg0, upsample=True)
            if i:
                t_c0.append(time.perf_counter() - t0)
    del sd0

    # -- headline workload: Swin-L, one 1024 x 2048 image, forward + backward, two samples
    cfg = T.ModelCfg(swin=T.SWIN_L)
    sd = {k: v.requires_grad_() for k, v in fill.state_dict_for(T.model_param_shapes(cfg)).items()}

    def step(h, w):
        img = torch.randint(0, 256, (3, h, w), generator=g).float()
        for v in sd.values():
            v.grad = None
        out = T.oneformer_forward([{"left_image": img, "task": "The task is panoptic"}], sd, cfg, upsample=True)
        T.synthetic_loss(out).backward()

    print(f"[bench] cpu_baseline: oracle on {cores} threads, Swin-L {H_IMG}x{W_IMG} fwd+bwd, one image x 2 samples ...", file=sys.stderr, flush=True)
    step(96, 192)                # warm-up (allocator, thread pool) on a small image
    samples = []
    for _ in range(2):
        t0 = time.perf_counter()
        step(H_IMG, W_IMG)
        samples.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline: sample {len(samples)}: {samples[-1]:.1f} s", file=sys.stderr, flush=True)
    dt = sum(samples) / len(samples)
    return {"value": round(1.0 / dt, 5), "unit": "img/s", "cores": cores, "kind": "port",
            "samples_s": [round(x, 2) for x in samples],
            "sample": f"oracle/torch_ref.py fp32 fwd+bwd (incl. the x4 mask upsample), full Swin-L OneFormer, ONE image {H_IMG}x{W_IMG} per sample, "
                      f"2 samples after a small-image warm-up ({samples[0]:.1f} s, {samples[1]:.1f} s) on {cores} threads (bs 1 of the metric's bs 2; no extrapolation)",
            "c0": {"value": round(1.0 / (sum(t_c0) / len(t_c0)), 4), "unit": "img/s (forward only)", "samples_s": [round(x, 3) for x in t_c0],
                   "workload": "BASELINE configs[0]: full OneFormer with the Swin-T backbone (C 96, depths 2-2-6-2, ws 7), one 512x1024 image, forward incl. the "
                               "x4 mask upsample, 1 warm-up + 3 timed"}}


def gpu_eager_baseline(device, steps=2):
    """The fp32 oracle's FORWARD on this GPU through ATen (eager PyTorch-ROCm): the "reference PyTorch path" of north_star's >= 3x
    target, bs 2 at 1024 x 2048, including the x4 mask upsample.  A baseline, never the product."""
    from oracle import fill, torch_ref as T
    cfg = T.ModelCfg(swin=T.SWIN_L)
    sd = {k: v.to(device) for k, v in fill.state_dict_for(T.model_param_shapes(cfg)).items()}
    g = torch.Generator().manual_seed(1000)
    imgs = [torch.randint(0, 256, (3, H_IMG, W_IMG), generator=g).float().to(device) for _ in range(PER_GPU_BATCH)]
    batch = [{"left_image": im, "task": "The task is panoptic"} for im in imgs]

    def fwd():
        with torch.no_grad(), torch.device(device):
            return T.oneformer_forward(batch, sd, cfg, upsample=True)
    fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(PER_GPU_BATCH / dt, 3), "unit": "img/s (forward only)", "ms_per_forward": round(dt * 1e3, 1),
            "what": "the BUILDER'S OWN restatement oracle/torch_ref.py (fp32, eager ATen ops on this GPU; written for clarity, not the "
                    "reference's code and not tuned), bs 2 1024x2048, forward + mask upsample"}


def _kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "uni-encoder-code_amd", "csrc")
    for n in sorted(os.listdir(d)):
        if n.endswith((".hip", ".h")):
            with open(os.path.join(d, n), "rb") as f:
                h.update(n.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def _quoted_profile(name):
    """A committed rocprofv3 summary (profiles/<round>_<name>.json) is quoted only if it was measured on the kernel sources
    this run executes (`kernel_sources_sha` recorded by the tool that wrote it); otherwise None, never a stale constant."""
    try:
        with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{name}.json")) as f:
            d = json.load(f)
        return d if d.get("kernel_sources_sha") == _kernel_sources_sha() else None
    except Exception:
        return None


def dp_check(model, buckets, batches, step_fn, rank, world, device):
    """N > 1 correctness of the real model's data-parallel path: the all-reduced gradients of one step (every rank on its own
    batch) against the mean of `world` single-rank backward passes on the same batches, computed locally by every rank."""
    from uenc import ops
    step_fn(batches[rank])
    torch.cuda.synchronize()
    reduced = buckets.flat.clone()
    saved_world, saved_coll = buckets.world, buckets._collective
    buckets.world, buckets._collective = 1, False      # local passes: no collective, no averaging
    acc = torch.zeros_like(reduced)
    for r in range(world):
        step_fn(batches[r])
        acc += buckets.flat
    acc /= world
    step_fn(batches[0]); a = buckets.flat.clone()
    step_fn(batches[0]); b = buckets.flat.clone()      # run-to-run noise floor (float atomics order)
    buckets.world, buckets._collective = saved_world, saved_coll
    torch.cuda.synchronize()
    per = []
    for i, v in buckets._views.items():
        s = v.data_ptr() - buckets.flat.data_ptr()
        s //= 4
        n = v.numel()
        ref = acc[s:s + n]
        den = float(ref.norm()) + 1e-30
        per.append((float((reduced[s:s + n] - ref).norm()) / den, float((a[s:s + n] - b[s:s + n]).norm()) / (float(a[s:s + n].norm()) + 1e-30), i))
    names = {id(p): n for n, p in model.named_parameters()}
    worst = sorted(per, reverse=True)[:5]
    rec = {"world": world, "rank": rank, "parameters_reduced": len(per), "buckets": len(buckets.bucket_ranges),
           "flat_mb": round(buckets.flat.numel() * 4 / 2 ** 20, 1),
           "global_rel": float((reduced - acc).norm() / acc.norm()), "global_run_to_run_rel": float((a - b).norm() / a.norm()),
           "max_param_rel": worst[0][0], "median_param_rel": statistics.median(x[0] for x in per),
           "max_param_run_to_run_rel": max(x[1] for x in per),
           "worst": [{"param": names.get(id(buckets.params[i]), str(i)), "rel": r, "run_to_run_rel": rr} for r, rr, i in worst],
           "what": "all-reduced gradients of one DP step vs the mean of `world` single-rank passes on the same batches; "
                   "differences = float-atomics order (see run_to_run_rel: two identical local passes)"}
    return rec


def _self_launch(n):
    """`python bench.py --gpus N` without torchrun: start the N ranks here, one process per GPU, the way the reference's
    detectron2.engine.launch does (train_net.py:302-309) -- as a child `torch.distributed.run` started BEFORE this process makes
    any GPU call (never an exec from a process that has initialised the GPU), and exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks through torch.distributed.run on 127.0.0.1:{port}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _rendezvous_only(args, world, rank):
    """Plumbing check of the N > 1 entry (CPU test, gloo): rendezvous, barrier, the MAX-over-ranks reduction of the timed region,
    rank 0's JSON line -- no model, no GPU, `value` null."""
    dist.init_process_group("gloo")
    dist.barrier()
    t0 = time.perf_counter()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "img/s fwd+bwd Swin-L 1024x2048 bs=2 per GPU", "value": None, "unit": "img/s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "rendezvous_only": True, "config": {"parallelism": f"dp{world}"}}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip forward-only / train-mode / eager-baseline legs (profiling runs)")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--check-dp", default="", help="N > 1: verify the reduced gradients against single-rank passes, write this JSON")
    ap.add_argument("--rendezvous-only", action="store_true", help="N ranks rendezvous over gloo and print the line's skeleton (CPU plumbing test)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))          # never report dp1 for --gpus N: start the ranks, or fail with the child's code
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (torch.distributed.run --nproc-per-node {args.gpus}, "
                         "or plain `python bench.py --gpus N`, which launches them itself)")
    if args.rendezvous_only:
        return _rendezvous_only(args, world, rank)
    if world > max(torch.cuda.device_count(), 1) and os.environ.get("UENC_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible (RCCL needs one device per rank; "
                         "UENC_DIST_BACKEND=gloo runs a functional rehearsal of more ranks than GPUs)")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    force_coll = world == 1 and os.environ.get("UENC_DP_FORCE_COLLECTIVE") == "1"      # one-GPU rehearsal of the RCCL call path (uenc/dp.py)
    if force_coll:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            free_port = so.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port))
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_coll:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("UENC_DIST_BACKEND", "nccl")      # "gloo": functional rehearsal of N ranks on fewer GPUs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    from uenc import capi, ops
    from uenc.d2 import build_model
    from uenc.dp import GradBuckets

    torch.manual_seed(0)                       # identical replicas (GradBuckets also broadcasts rank 0's parameters)
    model = build_model(make_cfg(device))
    model.eval()                               # deterministic path: dropout / stochastic depth = identity (the full FLOPs)
    if os.environ.get("UENC_BENCH_TRAIN") == "1":        # profiling aid: the timed steps themselves in training mode (workload string says so)
        model.train()
    buckets = GradBuckets(model, bucket_mb=args.bucket_mb)

    def make_batch(r):
        g = torch.Generator().manual_seed(1000 + r)
        return [{"left_image": torch.randint(0, 256, (3, H_IMG, W_IMG), generator=g).float().to(device),
                 "task": "The task is panoptic", "type": "segmentation", "height": H_IMG, "width": W_IMG}
                for _ in range(PER_GPU_BATCH)]
    batch = make_batch(rank)

    def step(b=None):
        buckets.zero_grad()
        ops.begin_step(fresh_grads=True)       # weights change every training step: every bf16 operand copy is re-cast; grads are zero
        out, images = model.forward_features(b if b is not None else batch)
        with torch.no_grad():                  # reference :255-263, part of the forward it times
            model.upsample_masks(out["pred_masks"], images.tensor.shape[-2:])
        loss = synthetic_loss(out)
        loss.backward()
        buckets.finish()
        return loss

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    sync()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        evs[i][0].record()
        loss = step()
        evs[i][1].record()
    sync()
    dt = time.perf_counter() - t0
    loss_val = float(loss.detach())
    step_ms = [a.elapsed_time(b) for a, b in evs]
    # host cost of one step, enqueued into an idle stream (the GPU is behind the host for the whole call)
    t1 = time.perf_counter()
    step()
    host_ms = (time.perf_counter() - t1) * 1e3
    sync()
    # per-launch timing of the GEMM families: HIP events recorded on the launch stream around every launch, over a repetition
    # of the same K steps -- inside the timed region the ~1200 event pairs per step cost 4 % of `value` (measured: 23.4 vs
    # 24.4 img/s), so the timed region itself stays un-instrumented
    capi.lib.uenc_prof_enable(1)
    for _ in range(args.steps):
        step()
    sync()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    fams, alg_bytes = {}, {}
    for kind, name in ((4, "gemm_nt256_kernel"), (5, "gemm_nt128_kernel"), (0, "gemm_nt_kernel"), (1, "gemm_tn*_kernel")):
        ms, fl, n, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        capi.lib.uenc_prof_collect(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        capi.lib.uenc_prof_collect_bytes(kind, ctypes.byref(by))
        fams[name] = (ms.value, fl.value, n.value)
        alg_bytes[name] = by.value
    capi.lib.uenc_prof_enable(0)

    dp_rec = None
    if world > 1 and args.check_dp:
        dp_rec = dp_check(model, buckets, [make_batch(r) for r in range(world)], step, rank, world, device)

    extras = {}
    if world == 1 and not args.no_extras:
        # forward only (inference): no_grad, incl. the x4 mask upsample
        def fwd():
            with torch.no_grad():
                out, images = model.forward_features(batch)
                model.upsample_masks(out["pred_masks"], images.tensor.shape[-2:])
        for _ in range(2):
            fwd()
        torch.cuda.synchronize()
        tf = time.perf_counter()
        nf = max(4, args.steps)
        for _ in range(nf):
            fwd()
        torch.cuda.synchronize()
        tf = (time.perf_counter() - tf) / nf
        extras["forward_only"] = {"value": round(PER_GPU_BATCH / tf, 3), "unit": "img/s", "ms_per_forward": round(tf * 1e3, 3)}
        # training mode: stochastic depth (Swin DropPath 0..0.3), dropout 0.1 of the deformable encoder and the class transformer
        model.train()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tt = time.perf_counter()
        nt = max(4, args.steps)
        for _ in range(nt):
            step()
        torch.cuda.synchronize()
        tt = (time.perf_counter() - tt) / nt
        model.eval()
        extras["train_mode"] = {"ms_per_step": round(tt * 1e3, 3), "value": round(PER_GPU_BATCH / tt, 3), "unit": "img/s",
                                "what": "model.train(): DropPath 0..0.3 on both residual branches of every Swin block (per-image scale inside the GEMM "
                                        "epilogues), dropout 0.1 in the deformable encoder layers (index-hash masks between the fused layer's kernels) "
                                        "and in the class transformer (attention-probability dropout inside the attention kernels)"}
        try:
            loss = None
            buckets.zero_grad()
            torch.cuda.empty_cache()
            eb = gpu_eager_baseline(device)
            eb["product_forward_speedup"] = round(extras["forward_only"]["value"] / eb["value"], 2)
            extras["gpu_eager_baseline"] = eb
        except Exception as e:                      # e.g. out of memory beside the product's buffers: report, do not fail the line
            extras["gpu_eager_baseline"] = {"error": repr(e)[:200]}

    if rank == 0:
        imgs = PER_GPU_BATCH * world * args.steps
        dom = max(fams, key=lambda k: fams[k][0])
        ms, fl, n = fams[dom]
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        gemm_ms = sum(v[0] for v in fams.values())
        # L2 <-> fabric bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process; the committed
        # summary of the two rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py) is quoted when it matches the sources
        traffic, traffic_src = None, None
        pmc = _quoted_profile("pmc_traffic")
        if pmc is not None:
            try:
                traffic = pmc["families"][dom]["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{PROFILE_ROUND}_pmc_traffic.json (rocprofv3 PMC passes at {pmc.get('git_head', '?')}, kernel sources {pmc['kernel_sources_sha']})"
            except Exception:
                pass
        mf = _quoted_profile("mfma_util")
        swin = BACKBONE != "dinat"
        rec = {
            "metric": "img/s fwd+bwd Swin-L 1024x2048 bs=2 per GPU" if swin else "img/s fwd+bwd DiNAT-L 1024x2048 bs=2 per GPU",
            "value": round(imgs / dt, 4), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: full OneFormer (Swin-L ws12 backbone + MSDeformAttn pixel decoder + "
                                    "150-query masked-attention decoder), 1024x2048, fwd+bwd, synthetic loss") if swin else
                                   ("BASELINE configs[4]: full OneFormer with the DiNAT-L backbone (neighbourhood attention, kernel 7), "
                                    "1024x2048, fwd+bwd, synthetic loss [UENC_BENCH_BACKBONE=dinat]"),
                       "global_batch": PER_GPU_BATCH * world, "parallelism": f"dp{world}", "mode": "train()" if os.environ.get("UENC_BENCH_TRAIN") == "1" else "eval() (train_mode reported separately)"},
            "step_ms": {"median": round(statistics.median(step_ms), 3), "min": round(min(step_ms), 3), "max": round(max(step_ms), 3)},
            "host_enqueue_ms": round(host_ms, 2),
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(achieved / 2500.0, 4), "traffic": traffic, "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(alg_bytes[dom] / max(n, 1)), "launches_per_step": n // max(args.steps, 1),
                         "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
                         "gemm_share_of_step": round(gemm_ms / (dt * 1e3), 3),
                         "families": {k: {"tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0.0,
                                          "ms_per_step": round(v[0] / max(args.steps, 1), 2), "launches_per_step": v[2] // max(args.steps, 1),
                                          "algorithmic_bytes_per_step": round(alg_bytes[k] / max(args.steps, 1)),
                                          "algorithmic_gbps": round(alg_bytes[k] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0.0}
                                      for k, v in fams.items()}},
            "loss": round(loss_val, 5),
        }
        if force_coll:
            rec["config"]["rccl_rehearsal"] = ("UENC_DP_FORCE_COLLECTIVE=1: every gradient bucket went through an RCCL all-reduce (ncclAvg, async) "
                                               "in a world of one rank -- the N > 1 call path on one GPU; not a scaling figure")
        if swin:
            rec["config"]["model_tflop_per_step_per_gpu"] = round(3 * GFLOP_FWD_PER_IMG * PER_GPU_BATCH / 1e3, 2)
            rec["model_tflops_per_gpu"] = round(3 * GFLOP_FWD_PER_IMG * PER_GPU_BATCH / 1e3 / (dt / args.steps), 1)
        if mf is not None:
            rec["mfma_util"] = {k: mf[k] for k in ("mfma_busy_frac_of_step", "mfma_busy_frac_of_gpu_busy", "source") if k in mf}
        rec.update(extras)
        if dp_rec is not None:
            rec["dp_check"] = dp_rec
            with open(args.check_dp, "w") as f:
                json.dump(dp_rec, f, indent=1)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
