"""Data-parallel gradient buckets over gloo, world size 2, on CPU (the N > 1 path of bench.py)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 8))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uenc.dp import GradBuckets
    m = _model()
    shared = m[2].weight                       # also used a second time below: a multi-contribution parameter
    gb = GradBuckets(m, bucket_mb=0.004, listen_ops=False)     # ~1k floats per bucket -> several buckets
    res = []
    for step in range(3):
        gb.zero_grad()
        g = torch.Generator().manual_seed(100 * step + rank)
        x = torch.randn(5, 16, generator=g)
        y = m(x)
        loss = y.square().mean() + shared.square().sum() * x.mean() * 1e-3
        loss.backward()
        gb.finish()
        res.append([p.grad.detach().clone().numpy() for p in m.parameters()])
        assert all(p.grad.data_ptr() >= gb.flat.data_ptr() for p in m.parameters())
    q.put((rank, res, len(gb.bucket_ranges)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_mean_of_local_grads():
    world, port = 2, 29541
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, nbuckets = {}, []
    try:
        for _ in range(world):
            r, res, nb = q.get(timeout=60)
            out[r] = res
            nbuckets.append(nb)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert all(nb >= 2 for nb in nbuckets)
    # single-process reference: average of the two ranks' local gradients
    m = _model()
    shared = m[2].weight
    for step in range(3):
        want = None
        for rank in range(world):
            m.zero_grad()
            g = torch.Generator().manual_seed(100 * step + rank)
            x = torch.randn(5, 16, generator=g)
            (m(x).square().mean() + shared.square().sum() * x.mean() * 1e-3).backward()
            gs = [p.grad.clone() for p in m.parameters()]
            want = gs if want is None else [a + b for a, b in zip(want, gs)]
        want = [w / world for w in want]
        for rank in range(world):
            for a, b in zip(out[rank][step], want):
                torch.testing.assert_close(torch.from_numpy(a), b, atol=1e-6, rtol=1e-5)


def test_single_process_buckets_track_signals():
    sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
    from uenc.dp import GradBuckets
    m = _model()
    gb = GradBuckets(m, bucket_mb=0.004, listen_ops=False)
    for _ in range(2):
        gb.zero_grad()
        m(torch.randn(4, 16)).sum().backward()
        gb.finish()
    assert gb._expected == [1] * len(gb.params) and not gb._calibrating
    # after calibration the layout follows backward order: the last layer's parameters come first
    first = gb.params[[i for i, b in gb._bucket_of.items() if b == 0][0]]
    assert any(first is p for p in m[4].parameters())
    gb.zero_grad()
    m(torch.randn(4, 16)).sum().backward()
    assert all(gb._launched) or gb.world == 1
    gb.finish()


def test_gradientless_parameters_are_excluded_and_grads_stay_in_the_flat_buffer():
    """Parameters that receive no gradient on the path (the reference builds depth / pose / motion decoders unconditionally,
    oneformer_model.py:143-145) leave the flat buffer after calibration; a .grad knocked out of the buffer is re-pointed by
    zero_grad() and caught by finish()."""
    sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
    from uenc.dp import GradBuckets

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.used = nn.Linear(8, 8)
            self.unused = nn.Linear(8, 1024)

        def forward(self, x):
            return self.used(x)
    m = M()
    gb = GradBuckets(m, bucket_mb=0.004, listen_ops=False)
    assert gb.flat.numel() == sum(p.numel() for p in m.parameters())
    gb.zero_grad(); m(torch.randn(3, 8)).sum().backward(); gb.finish()
    assert gb.flat.numel() == sum(p.numel() for p in m.used.parameters())          # excluded statically
    assert m.unused.weight.grad is None and m.unused.bias.grad is None
    want = m.used.weight.grad.clone()
    # an optimizer-style zero_grad(set_to_none=True) detaches .grad from the buffer: zero_grad() repairs it ...
    m.zero_grad(set_to_none=True)
    gb.zero_grad()
    assert m.used.weight.grad is not None and m.used.weight.grad.data_ptr() >= gb.flat.data_ptr()
    torch.manual_seed(0); x = torch.randn(3, 8)
    m(x).sum().backward(); gb.finish()
    assert float(gb.flat.abs().sum()) > 0
    # ... and finish() refuses to reduce a buffer the gradients did not go to
    gb.zero_grad()
    m.used.weight.grad = None
    m(x).sum().backward()
    with pytest.raises(RuntimeError):
        gb.finish()
    assert want.shape == m.used.weight.shape


def _diverging_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uenc.dp import GradBuckets
    torch.manual_seed(rank)                                   # different initial weights per rank: must be overwritten by rank 0's
    m = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 4))
    gb = GradBuckets(m, listen_ops=False)
    w0 = m[0].weight.detach().clone()
    gb.zero_grad()
    y = m[0](torch.ones(2, 4))
    if rank == 0:
        y = m[1](y)                                           # rank 1 skips a layer: a different graph
    y.sum().backward()
    try:
        gb.finish()
        q.put((rank, "no error", w0))
    except RuntimeError as e:
        q.put((rank, "raised" if "differ from rank 0 on rank(s) 1 " in str(e) else repr(e), w0))
    dist.destroy_process_group()


def test_replicas_are_broadcast_and_a_diverging_rank_is_caught():
    world, port = 2, 29547
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_diverging_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, what, w0 = q.get(timeout=60)
            got[r] = (what, w0)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert torch.equal(got[0][1], got[1][1])                  # parameters were broadcast from rank 0 at construction
    # BOTH ranks raise (the verdict is all-gathered): rank 0 must not carry on into a collective rank 1 never joins
    assert got[0][0] == "raised" and got[1][0] == "raised"


def _run_bench(args, env_extra=None, timeout=240):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts its two ranks itself (one process per GPU, as detectron2's launch does
    at reference train_net.py:302-309) and rank 0's line says n_gpus 2 -- never a silent dp1 run."""
    import json
    r = _run_bench(["--gpus", "2", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["rendezvous_only"] is True


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run_bench(["--gpus", "8", "--rendezvous-only"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
