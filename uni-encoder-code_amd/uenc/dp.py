"""Data parallelism over the GPUs of one node: one process per GPU, RCCL all-reduce of gradients.

Replaces the reference's `create_ddp_model(model, broadcast_buffers=False)` (tools/trainers/trainer.py:110,
torch DDP over NCCL).  The path shards by images only (LayerNorm / GroupNorm, no cross-sample op), so
the single exchange per step is the gradient all-reduce.

Design for MI355X / xGMI:
  * gradients live in a few large flat fp32 buckets (default 64 MiB) — the HIP wgrad kernels and
    autograd both accumulate straight into views of them (`p.grad`), so there is no copy-in/out and a
    bucket is one contiguous RCCL message (few, large collectives suit the per-link-bound xGMI mesh);
  * a parameter's gradient is *final* when it has received as many contributions as in a calibration
    step: the HIP Functions report each in-place accumulation (`ops.set_grad_listener`), autograd-
    produced gradients report through `register_post_accumulate_grad_hook`; no assumption about module
    boundaries or engine order is made;
  * buckets are laid out in the order gradients became final during calibration (true backward
    order), and a bucket's all-reduce is launched asynchronously (RCCL's stream) the moment its last
    gradient is final, overlapping the rest of backward;
  * parameters that never receive a gradient on this path (the depth / pose / motion decoders the reference builds
    unconditionally, oneformer_model.py:143-145) are dropped from the flat buffer after the calibration step: they are
    neither stored nor reduced;
  * replicas are made identical at construction (parameters and buffers broadcast from rank 0, as DDP does), and the bucket
    layout every rank uses is rank 0's: the observed order and contribution counts are broadcast after calibration and each
    rank checks that its own observation agrees, so ranks can never reduce different parameters against each other.
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class GradBuckets:
    def __init__(self, model: nn.Module, bucket_mb: float = 64.0, process_group=None, listen_ops: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cap = max(1, int(bucket_mb * (1 << 20) / 4))
        self.params: List[nn.Parameter] = [p for p in model.parameters() if p.requires_grad]
        if self.world > 1:                      # identical replicas, whatever each rank's initialisation did
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            staged = dist.get_backend(process_group) == "gloo"
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    if staged and t.is_cuda:            # gloo rehearsal on a GPU: through host memory
                        host = t.data.cpu()
                        dist.broadcast(host, src=src, group=process_group)
                        t.data.copy_(host)
                    else:
                        dist.broadcast(t.data, src=src, group=process_group)
        self._index: Dict[int, int] = {id(p): i for i, p in enumerate(self.params)}
        self._expected: Optional[List[int]] = None          # contributions per parameter per step
        self._count = [0] * len(self.params)
        self._order: List[int] = []                          # calibration: parameters in the order they became final
        self._calibrating = True
        self._layout(list(range(len(self.params))))
        self._works = []
        # the 1 / world of the gradient mean rides in the collective where the backend can do it (RCCL: ncclAvg), so no
        # separate pass over the flat buffer follows the last all-reduce; gloo (CPU tests, rehearsal) sums and scales
        self._avg_in_collective = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        # UENC_DP_FORCE_COLLECTIVE=1: issue the bucket all-reduces even in a world of one rank (the mean over one rank is the identity).
        # The only way to run the RCCL call path -- ncclAvg, async handles on views of the flat buffer, the wait -- on a one-GPU box.
        self._collective = self.world > 1 or (dist.is_initialized() and os.environ.get("UENC_DP_FORCE_COLLECTIVE") == "1")
        self._hooks = [p.register_post_accumulate_grad_hook(self._autograd_hook) for p in self.params]
        if listen_ops:
            from . import ops
            ops.set_grad_listener(self.signal)
            if self.world > 1:
                ops.WGRADS.overlap_exchange()

    # ---- flat storage -----------------------------------------------------------------------
    def _layout(self, order: List[int]):
        total = sum(self.params[i].numel() for i in order)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.bucket_ranges, self._bucket_of, self._bucket_pending0 = [], {}, []
        self._views: Dict[int, torch.Tensor] = {}
        off = start = 0
        members = 0
        for i in order:
            p = self.params[i]
            n = p.numel()
            self._views[i] = self.flat[off:off + n].view_as(p)
            p.grad = self._views[i]
            self._bucket_of[i] = len(self.bucket_ranges)
            off += n
            members += 1
            if off - start >= self.cap:
                self.bucket_ranges.append((start, off)); self._bucket_pending0.append(members)
                start, members = off, 0
        if off > start:
            self.bucket_ranges.append((start, off)); self._bucket_pending0.append(members)
        self._pending = list(self._bucket_pending0)
        self._launched = [False] * len(self.bucket_ranges)

    # ---- readiness signals --------------------------------------------------------------------
    def _autograd_hook(self, p):
        self.signal(p)

    def signal(self, p):
        """One gradient contribution has been accumulated into p.grad."""
        i = self._index.get(id(p))
        if i is None:
            return
        self._count[i] += 1
        if self._calibrating:
            if self._count[i] == 1:
                self._order.append(i)
            else:                       # final position = last contribution
                self._order.remove(i); self._order.append(i)
            return
        if self._count[i] == self._expected[i]:
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch(b)

    def _launch(self, b: int):
        self._launched[b] = True
        if self._collective:
            s, e = self.bucket_ranges[b]
            if self.flat.is_cuda and dist.get_backend(self.group) == "gloo":
                # functional rehearsal of N ranks on fewer GPUs (bench.py, UENC_DIST_BACKEND=gloo): the bucket is staged through
                # host memory, synchronously -- correctness of the signalling / bucketing, not a performance path
                host = self.flat[s:e].cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                self.flat[s:e].copy_(host)
                return
            if self._avg_in_collective:
                try:
                    self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                    return
                except (RuntimeError, ValueError, TypeError):       # a communicator library without ncclAvg: refused before anything is
                    self._avg_in_collective = False                 # enqueued, on every rank alike -> sum here, scale in finish()
            self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- step protocol --------------------------------------------------------------------------
    def zero_grad(self):
        """Start of a step: zero the flat buffer and re-point every .grad at its slice of it (a `zero_grad(set_to_none=True)`
        or an optimizer that replaced .grad would otherwise leave the kernels accumulating outside the reduced buffer)."""
        self.flat.zero_()
        for i, v in self._views.items():
            p = self.params[i]
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        try:
            from . import ops
            ops.WGRADS.reset()                  # a backward that raised leaves queued groups / held notifications behind
        except Exception:
            pass
        self._count = [0] * len(self.params)
        self._pending = list(self._bucket_pending0)
        self._launched = [False] * len(self.bucket_ranges)
        self._works = []
        if self._calibrating:
            self._order = []

    def finish(self):
        """After backward: reduce whatever has not been launched, wait, average.  The first call also
        fixes the bucket layout from the observed gradient order (gradients of this step are reduced first)."""
        for b in range(len(self.bucket_ranges)):
            if not self._launched[b]:
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []
        if self.world > 1 and not self._avg_in_collective:
            self.flat.mul_(1.0 / self.world)
        for i, v in self._views.items():         # the reduced buffer must be what the kernels accumulated into
            g = self.params[i].grad
            if g is None or g.data_ptr() != v.data_ptr():
                raise RuntimeError("GradBuckets: a parameter's .grad no longer aliases the flat all-reduce buffer "
                                   "(call GradBuckets.zero_grad(), not optimizer.zero_grad(set_to_none=True))")
        if self._calibrating:
            self._calibrating = False
            expected, order = list(self._count), list(self._order)
            if self.world > 1:
                # one layout for every rank: rank 0's observation is broadcast; a rank whose own backward produced a different
                # set of gradients has a different graph and must not silently reduce against the others
                box = [(expected, order)]
                dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                           group=self.group)
                # the verdict is collective: every rank learns whether ANY rank diverged and all of them raise together -- a rank
                # that carried on alone would block in its next all-reduce (RCCL has no timeout by default)
                bad = [i for i, (a, b) in enumerate(zip(box[0][0], expected)) if a != b]
                verdicts = [None] * self.world
                dist.all_gather_object(verdicts, (dist.get_rank(self.group), bad[:8]), group=self.group)
                diverged = [(r, b) for r, b in verdicts if b]
                if diverged:
                    raise RuntimeError("GradBuckets: gradient contributions differ from rank 0 on rank(s) "
                                       + ", ".join(f"{r} (parameters {b})" for r, b in diverged)
                                       + " -- replicas must run the same graph; every rank stops here")
                expected, order = box[0]
            self._expected = expected
            active = [i for i in order if expected[i] > 0]
            old = {i: self.params[i].grad.clone() for i in active}
            for i in range(len(self.params)):       # gradient-less parameters: excluded statically, neither stored nor reduced
                if expected[i] == 0:
                    self.params[i].grad = None
            self._layout(active)
            for i, gi in old.items():
                self.params[i].grad.copy_(gi)

    def close(self):
        for h in self._hooks:
            h.remove()
        try:
            from . import ops
            ops.set_grad_listener(None)
        except Exception:
            pass
