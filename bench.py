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
is the MAX over ranks; `value` = images all ranks processed / that time.

`roofline`: the dominant kernel family (bf16 MFMA GEMM) timed per launch with HIP events on its launch
stream (`uenc_prof_*`) over a repetition of the timed steps, algorithmic FLOPs = 2*M*N*K per launch.
`cpu_baseline`: the fp32 oracle (oracle/torch_ref.py, a "port") forward+backward on the host cores, on a
bounded sample (one image at reduced resolution), converted to the metric's unit by pixel count.
"""
import argparse
import ctypes
import json
import os
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


class _MeanSquare(torch.autograd.Function):
    """mean(x^2) as one reduction forward and one scaled copy backward (autograd's pow / mean chain is five
    full passes over every 157 MB mask tensor)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.linalg.vector_norm(x.float()).square() / x.numel()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return x * (g * (2.0 / x.numel())).to(x.dtype)


def synthetic_loss(out):
    """mean-square of logits and masks, x0.1 on the nine auxiliary predictions."""
    ms = _MeanSquare.apply
    loss = ms(out["pred_logits"]) + ms(out["pred_masks"])
    for a in out["aux_outputs"]:
        loss = loss + 0.1 * (ms(a["pred_logits"]) + ms(a["pred_masks"]))
    return loss


def cpu_baseline(budget_h=256, budget_w=512):
    """fp32 oracle forward+backward of the same Swin-L model on one (budget_h x budget_w) image on the host cores."""
    from oracle import fill, torch_ref as T
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)       # a 1-GPU box grants 16 CPUs whatever the host has: more threads only oversubscribe them
    torch.set_num_threads(cores)
    cfg = T.ModelCfg(swin=T.SWIN_L)
    sd = {k: v.requires_grad_() for k, v in fill.state_dict_for(T.model_param_shapes(cfg)).items()}
    g = torch.Generator().manual_seed(0)

    def step(h, w):
        img = torch.randint(0, 256, (3, h, w), generator=g).float()
        for v in sd.values():
            v.grad = None
        out = T.oneformer_forward([{"left_image": img, "task": "The task is panoptic"}], sd, cfg, upsample=True)
        T.synthetic_loss(out).backward()

    print(f"[bench] cpu_baseline: oracle on {cores} threads ...", file=sys.stderr, flush=True)
    step(96, 192)                # warm-up (allocator, thread pool) on a small image
    t0 = time.perf_counter()
    step(budget_h, budget_w)
    dt = time.perf_counter() - t0
    frac = (budget_h * budget_w) / float(H_IMG * W_IMG)
    return {"value": round(frac / dt, 5), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"oracle/torch_ref.py fp32 fwd+bwd, full Swin-L OneFormer, 1 image {budget_h}x{budget_w} "
                      f"({dt:.1f} s), scaled by pixel count x{1 / frac:.0f} to 1024x2048"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("UENC_DIST_BACKEND", "nccl")      # "gloo": functional rehearsal of N ranks on fewer GPUs
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    from uenc import capi, ops
    from uenc.d2 import build_model
    from uenc.dp import GradBuckets

    torch.manual_seed(0)                       # identical replicas
    model = build_model(make_cfg(device))
    model.eval()                               # deterministic path: dropout / stochastic depth = identity (same FLOPs)
    buckets = GradBuckets(model, bucket_mb=args.bucket_mb)
    g = torch.Generator().manual_seed(1000 + rank)
    batch = [{"left_image": torch.randint(0, 256, (3, H_IMG, W_IMG), generator=g).float().to(device),
              "task": "The task is panoptic", "type": "segmentation", "height": H_IMG, "width": W_IMG}
             for _ in range(PER_GPU_BATCH)]

    def step():
        buckets.zero_grad()
        ops.CACHE.refresh()                    # weights change every training step: every bf16 operand copy is re-cast
        out, images = model.forward_features(batch)
        with torch.no_grad():                  # reference :255-263, part of the forward it times
            model.upsample_masks(out["pred_masks"], images.tensor.shape[-2:])
        loss = synthetic_loss(out)
        loss.backward()
        buckets.finish()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    sync()
    # No HIP graph: the step is GPU-bound -- the host enqueues its ~2000 launches in ~50 ms, the GPU needs ~80 (tools/
    # cpu_bound_probe.py) -- so replaying a captured graph would not shorten it.
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    # per-launch timing of the GEMM families: HIP events recorded on the launch stream around every launch, over a repetition
    # of the same K steps -- inside the timed region the ~1200 event pairs per step cost 4 % of `value` (measured: 23.4 vs
    # 24.4 img/s), so the timed region itself stays un-instrumented
    capi.lib.uenc_prof_enable(1)
    for _ in range(args.steps):
        step()
    sync()
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    fams, alg_bytes = {}, {}
    for kind, name in ((4, "gemm_nt256_kernel"), (0, "gemm_nt_kernel"), (1, "gemm_tn*_kernel")):
        ms, fl, n, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        capi.lib.uenc_prof_collect(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
        capi.lib.uenc_prof_collect_bytes(kind, ctypes.byref(by))
        fams[name] = (ms.value, fl.value, n.value)
        alg_bytes[name] = by.value
    capi.lib.uenc_prof_enable(0)

    if rank == 0:
        imgs = PER_GPU_BATCH * world * args.steps
        dom = max(fams, key=lambda k: fams[k][0])
        ms, fl, n = fams[dom]
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        gemm_ms = sum(v[0] for v in fams.values())
        # L2 <-> fabric bytes per launch of the dominant kernel: PMC counters cannot be read from inside this process; the committed
        # summary of the two rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py) is quoted when present
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f)["families"][dom]["traffic_bytes_per_launch"]
        except Exception:
            pass
        rec = {
            "metric": "img/s fwd+bwd Swin-L 1024x2048 bs=2 per GPU" if BACKBONE != "dinat" else "img/s fwd+bwd DiNAT-L 1024x2048 bs=2 per GPU", "value": round(imgs / dt, 4), "unit": "img/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: full OneFormer (Swin-L ws12 backbone + MSDeformAttn pixel decoder + "
                                    "150-query masked-attention decoder), 1024x2048, fwd+bwd, synthetic loss") if BACKBONE != "dinat" else
                                   ("BASELINE configs[4]: full OneFormer with the DiNAT-L backbone (neighbourhood attention, kernel 7), "
                                    "1024x2048, fwd+bwd, synthetic loss [UENC_BENCH_BACKBONE=dinat; model_tflop fields do not apply]"),
                       "global_batch": PER_GPU_BATCH * world, "parallelism": f"dp{world}",
                       "model_tflop_per_step_per_gpu": round(3 * GFLOP_FWD_PER_IMG * PER_GPU_BATCH / 1e3, 2)},
            "model_tflops_per_gpu": round(3 * GFLOP_FWD_PER_IMG * PER_GPU_BATCH / 1e3 / (dt / args.steps), 1),
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(achieved / 2500.0, 4), "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 PMC pass, profiles/r01_pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": round(alg_bytes[dom] / max(n, 1)), "launches_per_step": n // max(args.steps, 1),
                         "avg_launch_us": round(ms * 1e3 / max(n, 1), 2),
                         "gemm_share_of_step": round(gemm_ms / (dt * 1e3), 3),
                         "families": {k: {"tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[0] > 0 else 0.0,
                                          "ms_per_step": round(v[0] / max(args.steps, 1), 2), "launches_per_step": v[2] // max(args.steps, 1)}
                                      for k, v in fams.items()}},
            "loss": round(float(loss.detach()), 5),
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
