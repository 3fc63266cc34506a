"""Autograd layer over the HIP kernels.

Conventions
  * master parameters are fp32 `nn.Parameter`s with the reference's names / shapes; the bf16 operand
    copies the MFMA kernels read are made by `CACHE` (re-cast whenever the parameter's version or
    storage changes, i.e. after every optimizer step);
  * activations between kernels are bf16, residual streams and everything LayerNorm reads are fp32;
  * weight / bias gradients are accumulated by the kernels straight into `param.grad` (fp32, created
    on demand), the way a fused gradient-accumulation does: the Functions return None for them.  A
    data-parallel wrapper (`uenc.dp`) can point `.grad` at flat all-reduce buckets beforehand.
"""
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import kernels as K

BF16, F32 = torch.bfloat16, torch.float32


# --------------------------------------------------------------------------------------------
# parameter operand cache and gradient buffers
# --------------------------------------------------------------------------------------------
class ParamCache:
    """bf16 (optionally transposed / zero-padded) copies of fp32 parameters, keyed by version.

    `refresh()` re-casts every cached copy in place with ONE batched kernel launch (instead of one tiny launch per
    tensor): call it after each optimizer step; entries are otherwise re-made lazily when a parameter's version or
    storage changed."""

    def __init__(self):
        self._store = {}
        self.casts = 0
        self._table = None          # (device descriptor tensor, n, total_tiles, keys)

    def invalidate(self):
        self._store.clear()
        self._table = None

    def _get(self, p: torch.Tensor, kind, make, plan=None):
        key = (id(p), kind)
        ent = self._store.get(key)
        # id() of a dead tensor can be reused by a new one (with the same storage address and version 0):
        # the weak reference tells a live owner from a recycled id
        if ent is not None and ent[3]() is p and ent[0] == p._version and ent[1] == p.data_ptr():
            return ent[2]
        t = make()
        self.casts += 1
        self._store[key] = (p._version, p.data_ptr(), t, weakref.ref(p), plan)
        self._table = None
        return t

    @staticmethod
    def _mat_view(p, rows):
        w = p.detach().reshape(p.shape[0], -1)
        return w if rows is None else w[rows[0]:rows[1]]

    def mat(self, p: torch.Tensor, rows: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """(N, K) bf16 view of a Linear / 1x1-conv weight, rows [a, b) if given; K, N padded to 8."""
        w = self._mat_view(p, rows)
        N, Kd = w.shape
        aligned = N % 8 == 0 and Kd % 8 == 0

        def make():
            ww = w if aligned else torch.nn.functional.pad(w, (0, -Kd % 8, 0, -N % 8))
            return K.cast_bf16(ww.contiguous())
        plan = ((rows[0] if rows else 0) * Kd * 4, N, Kd, 0) if (aligned and not K.EXACT) else None
        return self._get(p, ("m", rows), make, plan)

    def mat_t(self, p: torch.Tensor, rows: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        """(K, N) bf16: the transposed operand the dgrad GEMM reads."""
        w = self._mat_view(p, rows)
        N, Kd = w.shape
        aligned = N % 8 == 0 and Kd % 8 == 0

        def make():
            ww = w if aligned else torch.nn.functional.pad(w, (0, -Kd % 8, 0, -N % 8))
            return K.cast_transpose_bf16(ww.contiguous())
        plan = ((rows[0] if rows else 0) * Kd * 4, N, Kd, 1) if (aligned and not K.EXACT) else None
        return self._get(p, ("t", rows), make, plan)

    def _get2(self, p1, p2, kind, make, plan=None):
        """Like _get for an operand derived from two parameters (valid while neither changed).  plan: a LIST of
        (source byte pointer, destination element offset, rows, cols, transpose | ld << 4) pieces for the batched refresh."""
        key = (id(p1), (kind, id(p2)))
        ent = self._store.get(key)
        sig = (p1._version, p1.data_ptr(), p2._version, p2.data_ptr())
        if ent is not None and ent[3]() is p1 and ent[5]() is p2 and ent[0] == sig:
            return ent[2]
        t = make()
        self.casts += 1
        self._store[key] = (sig, p1.data_ptr(), t, weakref.ref(p1), plan, weakref.ref(p2))
        self._table = None
        return t

    def cat(self, p1: torch.Tensor, p2: torch.Tensor, transposed: bool = False) -> torch.Tensor:
        """bf16 [p1; p2] stacked along the output dimension ((N1+N2, K), or its (K, N1+N2) transpose): two Linear layers that
        read the same input run as one GEMM.  Refreshed by the batched cast like any single weight: the two masters are two
        descriptors that fill row (column) blocks of the one destination -- no torch.cat, no cast launch of its own per step."""
        N1, N2 = p1.shape[0], p2.shape[0]
        Kd = p1.numel() // N1

        def make():
            w = torch.cat([p1.detach().reshape(N1, -1), p2.detach().reshape(N2, -1)], 0).contiguous()
            return K.cast_transpose_bf16(w) if transposed else K.cast_bf16(w)
        plan = None
        if (not K.EXACT and p1.is_contiguous() and p2.is_contiguous() and p2.numel() // N2 == Kd and Kd % 8 == 0 and N1 % 8 == 0 and N2 % 8 == 0
                and p1.dtype == F32 and p2.dtype == F32):
            ld = N1 + N2
            plan = ([(p1.data_ptr(), 0, N1, Kd, 1 | (ld << 4)), (p2.data_ptr(), N1, N2, Kd, 1 | (ld << 4))] if transposed
                    else [(p1.data_ptr(), 0, N1, Kd, 0), (p2.data_ptr(), N1 * Kd, N2, Kd, 0)])
        return self._get2(p1, p2, "catT" if transposed else "cat", make, plan)

    def catvec(self, p1: torch.Tensor, p2: torch.Tensor) -> torch.Tensor:
        return self._get2(p1, p2, "catv", lambda: torch.cat([p1.detach(), p2.detach()]).float().contiguous())

    def vec16(self, p: torch.Tensor) -> torch.Tensor:
        return self._get(p, "v", lambda: K.cast_bf16(p.detach().contiguous()), None if K.EXACT else (0, 1, p.numel(), 0))

    def relpos(self, table: torch.Tensor, ws: int) -> torch.Tensor:
        """Expanded relative-position bias (nH, NP, NP) fp32 of a window-attention module (the kernels' `bias_q`; query-major, scaled by
        log2(e)), kept while the table is unchanged; `refresh()` re-expands the biases of ALL modules with one grouped launch (24
        launches of ~9 us per Swin-L step otherwise).  Verification mode: the fp32 kernels index the table themselves."""
        if K.EXACT:
            return table.detach()
        ok = table.dtype == F32 and table.is_contiguous()
        return self._get(table, ("relpos", ws), lambda: K.relpos_expand(table.detach().contiguous(), ws)[0],
                         ("relpos", ws, table.shape[1]) if ok else None)

    def refresh(self):
        """Re-cast every cached operand copy from its (updated) fp32 master, in one launch."""
        import numpy as np
        from .capi import check, lib, stream_ptr
        WGRADS.reset()                              # start of a step: nothing of an earlier (failed) backward may survive
        if not self._store:
            return
        dead = [k for k, e in self._store.items() if e[3]() is None or e[4] is None or e[3]().data_ptr() != e[1]
                or (len(e) > 5 and (e[5]() is None or e[5]().data_ptr() != e[0][3]))]
        for k in dead:                              # padded / special entries and moved storages are re-made lazily
            del self._store[k]
            self._table = None
        if not self._store:
            return
        if self._table is None:
            pieces, keys, dev = [], [], None
            rel = []                                            # (table ptr, bias ptr, nH, ws, NP) of the expanded relative-position biases
            for k, e in self._store.items():
                if isinstance(e[4], tuple) and e[4] and e[4][0] == "relpos":
                    rel.append((e[1], e[2].data_ptr(), e[4][2], e[4][1], e[2].shape[-1]))
                    keys.append(k)
                    dev = e[2].device
                    continue
                if isinstance(e[4], list):                      # an operand stacked from two masters: one piece per master
                    for src, doff, rows, cols, tr in e[4]:
                        pieces.append((src, e[2].data_ptr() + 2 * doff, rows, cols, tr))
                else:
                    off, rows, cols, tr = e[4]
                    pieces.append((e[1] + off, e[2].data_ptr(), rows, cols, tr))
                keys.append(k)
                dev = e[2].device
            desc = np.zeros(len(pieces), dtype=[("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4"),
                                                ("tr", "<i4"), ("tc", "<i4"), ("tb", "<i8")])
            tb = 0
            for i, (src, dst, rows, cols, tr) in enumerate(pieces):
                desc[i] = (src, dst, rows, cols, tr, -(-cols // 64), tb)
                tb += -(-rows // 64) * -(-cols // 64)
            tab = torch.from_numpy(desc.view(np.uint8).copy()).to(dev) if len(pieces) else None
            rtab, rblocks = None, 0
            if rel:
                rdesc = np.zeros(len(rel), dtype=[("table", "<u8"), ("bias_q", "<u8"), ("bias_k", "<u8"), ("nH", "<i4"), ("ws", "<i4"),
                                                  ("NP", "<i4"), ("blk_begin", "<i4")])
                for i, (tp, bp, nH, ws, NP) in enumerate(rel):
                    rdesc[i] = (tp, bp, 0, nH, ws, NP, rblocks)
                    rblocks += -(-(nH * NP * NP) // 2048)
                rtab = torch.from_numpy(rdesc.view(np.uint8).copy()).to(dev)
            keys = (keys, len(pieces))
            self._table = (tab, keys[1], tb, keys[0], rtab, len(rel), rblocks)
        tab, n, total, keys, rtab, rn, rblocks = self._table
        if n:
            check(lib.uenc_cast_multi(tab.data_ptr(), n, total, stream_ptr()), "cast_multi")
        if rn:
            check(lib.uenc_relpos_expand_grouped(rtab.data_ptr(), rn, rblocks, stream_ptr()), "relpos_expand_grouped")
        for k in keys:
            e = self._store[k]
            if len(e) > 5:
                p1, p2 = e[3](), e[5]()
                self._store[k] = ((p1._version, p1.data_ptr(), p2._version, p2.data_ptr()), e[1], e[2], e[3], e[4], e[5])
            else:
                self._store[k] = (e[3]()._version, e[1], e[2], e[3], e[4])


CACHE = ParamCache()


def set_exact(flag: bool):
    """Switch the fp32 "exact" arithmetic mode (csrc/exact.hip) on or off for everything launched afterwards: fp32 operands and
    activations end to end.  A verification mode -- the product's mode is bf16 MFMA operands."""
    K.EXACT = bool(flag)
    CACHE.invalidate()
    _TWINS.clear()


def is_exact() -> bool:
    return K.EXACT


_GRAD_LISTENER = None


def set_grad_listener(fn):
    """`fn(param)` is called after each in-place accumulation into `param.grad` by a HIP Function
    (the data-parallel bucket scheduler uses it to know when a gradient is final)."""
    global _GRAD_LISTENER
    _GRAD_LISTENER = fn


def _notify(*params):
    if _GRAD_LISTENER is not None:
        for p in params:
            if p is not None and p.requires_grad:
                _GRAD_LISTENER(p)


def grad_buf(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def _pad_cols(x2: torch.Tensor, mult: int = 8) -> torch.Tensor:
    Kd = x2.shape[1]
    Kp = -(-Kd // mult) * mult
    return x2 if Kp == Kd else torch.nn.functional.pad(x2, (0, Kp - Kd))


def _bias_pad(b: Optional[torch.Tensor], Np: int) -> Optional[torch.Tensor]:
    if b is None or b.numel() == Np:
        return None if b is None else b.detach()
    return torch.nn.functional.pad(b.detach(), (0, Np - b.numel()))


class WgradQueue:
    """Deferred weight-gradient GEMMs, launched as groups.

    A layer's backward leaves a few wgrad GEMMs whose outputs are small (36 tiles of 256 x 256 for a 3072 x 768 weight):
    alone, each must split its token range ~7-way to occupy 256 CUs and pays an atomic burst and a pipeline fill per split.
    Nothing downstream in the backward needs dW, so the GEMMs are queued (their bf16 operands stay alive in HBM) and
    launched together -- one `uenc_gemm_tn_grouped` per tile class -- once enough work items for a few full waves of
    workgroups have gathered, and at the end of the backward pass (an autograd engine callback).  The gradient-ready
    notifications of the data-parallel bucket scheduler are held back until the group that contains them is launched.
    """
    _DESC = [("dy", "<u8"), ("x", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("ldy", "<i8"), ("ldx", "<i8"), ("ldw", "<i8"),
             ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("tiles_k", "<i4"), ("mlen", "<i4"), ("nsplit", "<i4"),
             ("item_begin", "<i4"), ("store", "<i4"), ("alpha", "<f4"), ("pad", "<i4")]
    _DESC_SMALL = [("dy", "<u8"), ("x", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("ldy", "<i8"), ("ldx", "<i8"), ("ldw", "<i8"),
                   ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("dy_f32", "<i4"), ("x_f32", "<i4"), ("tiles_k", "<i4"),
                   ("mlen", "<i4"), ("nsplit", "<i4"), ("item_begin", "<i4"), ("alpha", "<f4")]
    # measured on the Swin-L problem sets (tools/wgrad_group_bench.py): many short items beat few long ones (balance, more loads in
    # flight) until the per-item atomic burst shows, around 4k-8k tokens
    # `fresh` (set by begin_step): every gradient buffer is known to be zero at the start of this step, so the FIRST weight gradient
    # written to a buffer whose token range is a single item may be STORED instead of added -- plain stores run at ~5 TB/s, the
    # 256 KB-per-tile float-atomic bursts of the epilogue at the chip's 1.3 TB/s.  Bias gradients stay accumulated (other kernels
    # add to them too, e.g. the window-attention backward's padding-slot share of qkv.bias).
    fresh = False
    FLUSH_ITEMS = {256: 4096, 128: 16384}        # ~16 / 32 waves of workgroups per launch (bigger groups measured a little faster;
                                                 # flushing only at the end of backward would stall the gradient all-reduce overlap)
    TOKENS_PER_ITEM = {256: 16384, 128: 4096}     # token range of one work item (128 / 64 k-steps of 64)

    def __init__(self):
        ev = os.environ.get("UENC_WGRAD_FLUSH")
        if ev:                                   # A/B knob: items of the 256-tile class per launch (the 128-tile class gets 4x)
            self.FLUSH_ITEMS = {256: int(ev), 128: 4 * int(ev)}
        self.pending = {256: [], 128: []}        # tile -> [(desc tuple without item_begin, items, keepalive)]
        self.items = {256: 0, 128: 0}
        self.small = []                          # descriptors for the register-staged kernel (small / fp32-operand problems)
        self.small_items = 0
        self.notify = []
        self.callback_armed = False
        self.enabled = True
        self.written = set()                     # data_ptr of gradient buffers that received a contribution this step
        self.stores = {}                         # data_ptr -> store flag of a queued, not yet launched "store" descriptor

    def overlap_exchange(self):
        """Data parallel: launch smaller groups (about 4 waves of workgroups), so that the backbone's weight gradients become final
        -- and their buckets' all-reduce starts -- stage by stage instead of in one group at the end of the backward pass."""
        if not os.environ.get("UENC_WGRAD_FLUSH"):
            self.FLUSH_ITEMS = {256: 1024, 128: 4096}

    @staticmethod
    def eligible(dy, x, gw) -> bool:
        M = dy.shape[0]
        return (dy.dtype == BF16 and x.dtype == BF16 and dy.is_cuda and M % 64 == 0 and M >= 2048 and x.shape[0] == M
                and dy.stride(1) == 1 and x.stride(1) == 1 and gw.stride(1) == 1
                and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0
                and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and gw.dtype == F32)

    @staticmethod
    def eligible_small(dy, x, gw) -> bool:
        def ok(t):
            return (t.dim() == 2 and t.is_cuda and t.stride(1) == 1 and t.data_ptr() % 16 == 0 and
                    ((t.dtype == BF16 and t.stride(0) % 8 == 0) or (t.dtype == F32 and t.stride(0) % 4 == 0)))
        return (ok(dy) and ok(x) and x.shape[0] == dy.shape[0] and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0
                and gw.dtype == F32 and gw.stride(1) == 1 and gw.shape == (dy.shape[1], x.shape[1]))

    def add_small(self, dy, x, gw, gb, notify=(), alpha: float = 1.0):
        """Queue a problem for the grouped register-staged kernel (launched with the others at flush time)."""
        M, N, Kd = dy.shape[0], dy.shape[1], x.shape[1]
        mt = -(-M // 64)
        nsplit = max(1, min(mt // 8, -(-M // 2048)))
        mlen = -(-mt // nsplit) * 64
        nsplit = -(-M // mlen)
        tiles_k = -(-Kd // 128)
        items = -(-N // 128) * tiles_k * nsplit
        self.small.append(((dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else 0,
                            dy.stride(0), x.stride(0), gw.stride(0), M, N, Kd, int(dy.dtype == F32), int(x.dtype == F32), tiles_k, mlen,
                            nsplit), items, (dy, x, gw, gb), float(alpha)))
        self.small_items += items
        self.notify.extend(p for p in notify if p is not None)
        self._arm()
        if not self.callback_armed or self.small_items >= 4096:
            self.flush()

    def _arm(self):
        if not self.callback_armed:
            try:                                  # only valid while the autograd engine is running a backward pass
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self.callback_armed = True
            except RuntimeError:
                pass

    def busy(self) -> bool:
        return bool(self.notify or self.items[256] or self.items[128] or self.small_items or SMALLQ)

    def add(self, dy, x, gw, gb, notify=(), first=False, alpha: float = 1.0):
        M, N, Kd = dy.shape[0], dy.shape[1], x.shape[1]
        # 256 x 256 tiles re-read the operands half as often as 128 x 128 ones; they win unless padding N, K up to 256 wastes too much
        pad = lambda t: (-(-N // t) * t) * (-(-Kd // t) * t)
        tile = 256 if pad(256) <= 1.3 * pad(128) else 128
        nsplit = max(1, -(-M // self.TOKENS_PER_ITEM[tile]))
        mlen = -(-(M // 64) // nsplit) * 64
        nsplit = -(-M // mlen)
        tiles_k = -(-Kd // tile)
        items = -(-N // tile) * tiles_k * nsplit
        store = [2 if (self.fresh and nsplit == 1 and first) else 0]      # mutable: a later contribution to the same buffer clears it
        if store[0]:
            self.stores[gw.data_ptr()] = store
        self.pending[tile].append(((dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else 0,
                                    dy.stride(0), x.stride(0), gw.stride(0), M, N, Kd, tiles_k, mlen, nsplit), items, (dy, x, gw, gb), store,
                                   float(alpha)))
        self.items[tile] += items
        self.notify.extend(p for p in notify if p is not None)
        self._arm()
        if not self.callback_armed or self.items[tile] >= self.FLUSH_ITEMS[tile]:
            self.flush()

    @classmethod
    def launch(cls, tile: int, descs, device):
        """descs: [(dy_ptr, x_ptr, dw_ptr, db_ptr, ldy, ldx, ldw, M, N, K, tiles_k, mlen, nsplit, items, store[, alpha])]."""
        import numpy as np
        from .capi import check, lib, stream_ptr
        desc = np.zeros(len(descs), dtype=cls._DESC)
        begin, flops, nbytes = 0, 0.0, 0.0
        for i, d in enumerate(descs):
            desc[i] = d[:13] + (begin, d[14], d[15] if len(d) > 15 else 1.0, 0)
            begin += d[13]
            flops += 2.0 * d[7] * d[8] * d[9]
            nbytes += 2.0 * d[7] * (d[8] + d[9]) + 4.0 * d[8] * d[9]           # bf16 operands read once + the fp32 gradient written once
        host = torch.from_numpy(desc.view(np.uint8)).pin_memory()
        tab = host.to(device, non_blocking=True)
        lib.uenc_prof_next_bytes(nbytes)
        check(lib.uenc_gemm_tn_grouped(tab.data_ptr(), len(descs), begin, tile, flops, stream_ptr()), "gemm_tn_grouped")

    def _end_of_backward(self):
        self.callback_armed = False
        self.flush()

    def reset(self):
        """Drop whatever a previous, FAILED backward left queued (the engine skips its final callbacks when a node raises, so
        `callback_armed` would stay set and the stale groups would be launched -- a step late -- with the next pass)."""
        self.pending = {256: [], 128: []}
        self.items = {256: 0, 128: 0}
        self.small, self.small_items, self.notify = [], 0, []
        SMALLQ.clear()
        self.callback_armed = False
        self.written = set()
        self.stores = {}
        self.fresh = False

    def touch(self, gw) -> bool:
        """Record a contribution to gradient buffer `gw`; returns True if it is the first of this step.  A queued STORE into the same
        buffer is turned back into an accumulation (it will run after whatever is written now)."""
        key = gw.data_ptr()
        first = key not in self.written
        self.written.add(key)
        st = self.stores.pop(key, None)
        if st is not None:
            st[0] = 0
        return first

    def flush(self):
        """Launch everything queued (both tile classes), then release the held gradient-ready notifications."""
        self.stores = {}
        for tile in (256, 128):
            ent = self.pending[tile]
            if not ent:
                continue
            ent.sort(key=lambda e: -e[0][11])                       # longest token ranges first
            self.launch(tile, [d + (items, st[0], al) for d, items, _, st, al in ent], ent[0][2][0].device)
            self.pending[tile] = []
            self.items[tile] = 0
        if self.small:
            import numpy as np
            from .capi import check, lib, stream_ptr
            desc = np.zeros(len(self.small), dtype=self._DESC_SMALL)
            begin, flops = 0, 0.0
            for i, (d, items, _, al) in enumerate(self.small):
                desc[i] = d + (begin, al)
                begin += items
                flops += 2.0 * d[7] * d[8] * d[9]
            dev = self.small[0][2][0].device
            tab = torch.from_numpy(desc.view(np.uint8)).pin_memory().to(dev, non_blocking=True)
            check(lib.uenc_gemm_tn_grouped_small(tab.data_ptr(), len(self.small), begin, flops, stream_ptr()), "gemm_tn_grouped_small")
            self.small = []
            self.small_items = 0
        if SMALLQ:
            SMALLQ.flush()                      # parked LayerNorm / position-table partial sums: one grouped launch per kind
        if self.notify:
            params, self.notify = self.notify, []
            _notify(*params)


WGRADS = WgradQueue()

# Small parameter-gradient reductions (LayerNorm dgamma / dbeta block partials, relative-position-table partials) parked during the backward
# pass and summed by one grouped launch per kind when the weight-gradient queue is flushed (K.SmallReductions).  UENC_DEFER_SMALL=0: every
# pass reduces its own partials, as before (A/B).
SMALLQ = K.SmallReductions()
_DEFER_SMALL = os.environ.get("UENC_DEFER_SMALL", "1") != "0"


def _defer():
    return SMALLQ if (_DEFER_SMALL and WGRADS.enabled and not K.EXACT) else None


def _after_deferred():
    """A reduction was parked: make sure the end-of-backward flush runs; outside a backward pass (no engine callback) flush right away."""
    if SMALLQ:
        WGRADS._arm()
        if not WGRADS.callback_armed:
            WGRADS.flush()


def _ln_bwd(*a, **k):
    k.setdefault("defer", _defer())
    r = K.layernorm_bwd(*a, **k)
    _after_deferred()
    return r


def _pm_ln_bwd(*a, **k):
    k.setdefault("defer", _defer())
    r = K.patch_merge_ln_bwd(*a, **k)
    _after_deferred()
    return r


def _wattn_bwd(*a, **k):
    k.setdefault("defer", _defer() if k.get("dtable") is not None else None)
    r = K.window_attn_bwd(*a, **k)
    _after_deferred()
    return r


def begin_step(fresh_grads: bool = False):
    """Start of a training step (after the gradients were zeroed / re-pointed, before the forward): drops whatever a failed backward
    left queued and re-casts the bf16 operand copies of the updated weights.  fresh_grads=True declares that every gradient buffer
    is ZERO now (as after GradBuckets.zero_grad() or optimizer.zero_grad()): the first large weight gradient written to a buffer
    may then be stored instead of accumulated.  Leave it False when gradients are carried over from earlier backward passes."""
    CACHE.refresh()
    WGRADS.fresh = bool(fresh_grads)


def flush_wgrads():
    """Launch any weight-gradient GEMMs still queued (called automatically at the end of every backward pass)."""
    WGRADS.flush()


def _tn_notify(*params):
    """Gradient-ready notification that keeps its place behind queued wgrad groups."""
    if WGRADS.busy():
        WGRADS.notify.extend(p for p in params if p is not None)
    else:
        _notify(*params)


def _tn(dy: torch.Tensor, x: torch.Tensor, gw: torch.Tensor, gb: Optional[torch.Tensor], notify=(), alpha: float = 1.0):
    """gw += alpha * dy^T x, gb += alpha * column sums of dy: deferred to a grouped launch when the operands allow it."""
    first = WGRADS.touch(gw)
    if WGRADS.enabled and not K.EXACT and WGRADS.eligible(dy, x, gw):
        WGRADS.add(dy, x, gw, gb, notify, first, alpha)
        return
    if WGRADS.enabled and not K.EXACT and WGRADS.eligible_small(dy, x, gw):
        WGRADS.add_small(dy, x, gw, gb, notify, alpha)
        return
    K.gemm_tn(dy, x, gw, gb, alpha=alpha)
    if WGRADS.busy():
        WGRADS.notify.extend(p for p in notify if p is not None)     # keep notification order behind the queued groups
    else:
        _notify(*notify)


_BIG_M = 8192      # from this many rows on, an fp32 operand is first copied to bf16: the LDS-DMA GEMM kernels read bf16 only


# bf16 twins: producers that stream an fp32 tensor out (LayerNorm forward / backward) can write its bf16 copy in the same
# pass; consumers that need a bf16 GEMM operand ask here before launching a cast kernel.  Keyed by storage address, valid
# while the producing tensor object lives and its version counter (shared with every view) has not moved -- an in-place
# accumulation by the autograd engine into the fp32 tensor therefore drops the twin instead of leaving it stale.
_TWINS = {}


def _register_twin(t32: torch.Tensor, t16: torch.Tensor):
    if t16 is t32 or K.EXACT:               # exact mode: an fp32 tensor is its own operand copy (and must not keep itself alive here)
        return
    key = t32.data_ptr()

    def _gone(_, key=key):
        _TWINS.pop(key, None)
    _TWINS[key] = (weakref.ref(t32, _gone), t32._version, t32.numel(), t16)


def _twin(t: torch.Tensor) -> Optional[torch.Tensor]:
    e = _TWINS.get(t.data_ptr())
    if e is None or t.dtype != F32 or not t.is_contiguous():
        return None
    owner = e[0]()
    if owner is None or owner._version != e[1] or t._version != e[1] or t.numel() != e[2]:
        return None
    return e[3].view(t.shape)


def _as_bf16_operand(t2: torch.Tensor) -> torch.Tensor:
    if t2.dtype == F32 and t2.shape[0] >= _BIG_M and t2.is_contiguous() and t2.numel() % 8 == 0:
        tw = _twin(t2)
        if tw is None:
            tw = K.cast_bf16(t2)
            _register_twin(t2, tw)          # the same activation often feeds several GEMMs (k / v projections of every decoder
        return tw                           # layer, and the weight-gradient GEMM of each of them in the backward)
    return t2


def _wgrad(dy2: torch.Tensor, x2: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], rows=None):
    """Accumulate dW (+ db) of out = x W^T + b into the parameters' .grad."""
    if not w.requires_grad and (b is None or not b.requires_grad):
        return
    dy2, x2 = _as_bf16_operand(dy2), _as_bf16_operand(x2)
    N, Kd = dy2.shape[1], x2.shape[1]
    gw = grad_buf(w).view(w.shape[0], -1)
    if rows is not None:
        gw = gw[rows[0]:rows[1]]
    gb = None
    if b is not None and b.requires_grad:
        gb = grad_buf(b)
        if rows is not None:
            gb = gb[rows[0]:rows[1]]
    if N % 8 == 0 and Kd % 8 == 0 and gw.shape == (N, Kd):
        _tn(dy2, x2, gw, gb, (w, b))
        return
    # odd-sized tiny layers (class_embed N=20, task_mlp K=77): zero-padded scratch, then add the slice
    Np, Kp = -(-N // 8) * 8, -(-Kd // 8) * 8
    dyp, xp = _pad_cols(dy2), _pad_cols(x2)
    tw = torch.zeros((Np, Kp), dtype=F32, device=dy2.device)
    tb = torch.zeros((Np,), dtype=F32, device=dy2.device) if gb is not None else None
    K.gemm_tn(dyp.contiguous(), xp.contiguous(), tw, tb)
    gw.add_(tw[: gw.shape[0], : gw.shape[1]])
    if gb is not None:
        gb.add_(tb[: gb.numel()])
    _notify(w, b)


def _fwd_gemm(x2, w, b, rows, **kw):
    """x2 (M, K) @ W[rows]^T + b -> (M, N) with zero padding of odd K / N handled here."""
    w16 = CACHE.mat(w, rows)
    N = (rows[1] - rows[0]) if rows is not None else w.shape[0]
    bb = b.detach()[rows[0]:rows[1]] if (b is not None and rows is not None) else (b.detach() if b is not None else None)
    xp = _as_bf16_operand(_pad_cols(x2))
    if w16.shape[0] != N:
        bb = _bias_pad(bb, w16.shape[0])
    out = K.gemm_nt(xp, w16, bias=bb, **kw)
    return out if w16.shape[0] == N else out[:, :N]


def _dgrad_gemm(dy2, w, rows, **kw):
    wt = CACHE.mat_t(w, rows)                     # (Kp, Np)
    Kd = w.reshape(w.shape[0], -1).shape[1]
    dyp = _as_bf16_operand(_pad_cols(dy2))
    out = K.gemm_nt(dyp if dyp.shape[1] == wt.shape[1] else torch.nn.functional.pad(dyp, (0, wt.shape[1] - dyp.shape[1])),
                    wt, **kw)
    return out if wt.shape[0] == Kd else out[:, :Kd]


# --------------------------------------------------------------------------------------------
# Linear (optionally + fp32 residual), MLP chains
# --------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, rows, out_dtype):
        Kd = x.shape[-1]
        x2 = x.reshape(-1, Kd)
        if x2.stride(1) != 1 or (x2.stride(0) % 8) != 0:
            x2 = x2.contiguous()
        N = (rows[1] - rows[0]) if rows is not None else weight.shape[0]
        if residual is not None:
            r2 = residual.reshape(-1, N)
            # x W^T + b + residual is an fp32 stream by default; a caller whose sum only feeds further GEMMs may ask for bf16
            # (the 256-tile kernel's residual epilogue then rounds once on the way out: no fp32 round trip, no cast kernel)
            Md, Kd2 = x2.shape
            res16 = (out_dtype == BF16 and not K.EXACT and Md >= _BIG_M and N % 256 == 0 and Kd2 % 64 == 0 and Kd2 >= 128
                     and -(-Md // 256) * (N // 256) >= 160)
            out = _fwd_gemm(x2, weight, bias, rows, epilogue=K.EPI_RESIDUAL, aux=r2.contiguous(), out_dtype=BF16 if res16 else F32)
        else:
            out = _fwd_gemm(x2, weight, bias, rows, out_dtype=out_dtype)
        ctx.save_for_backward(x2, weight, bias)
        ctx.rows, ctx.has_res, ctx.xshape, ctx.xdtype = rows, residual is not None, x.shape, x.dtype
        return out.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        N = dy.shape[-1]
        dy2 = dy.reshape(-1, N)
        if dy2.stride(1) != 1 or (dy2.stride(0) % 8) != 0:
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad_gemm(dy2, weight, ctx.rows, out_dtype=ctx.xdtype).reshape(ctx.xshape)
        _wgrad(dy2, x2, weight, bias, ctx.rows)
        return dx, None, None, (dy if ctx.has_res else None), None, None


def linear(x, weight, bias=None, *, residual=None, rows=None, out_dtype=None):
    """y = x W[rows]^T + b[rows] (+ residual).  weight may be a conv 1x1 kernel (N, K, 1, 1).  out_dtype: bf16 by default; with a
    residual the result is fp32 unless bf16 is asked for explicitly (and the shape takes the 256-tile kernel)."""
    if out_dtype is None:
        out_dtype = F32 if residual is not None else BF16
    return LinearFn.apply(x, weight, bias, residual, rows, out_dtype)


class MLPFn(torch.autograd.Function):
    """x -> [Linear -> act] x (n-1) -> Linear (+ residual).  act in {"relu", "gelu"}.

    Backward fuses act' into the dgrad GEMM epilogue of the following layer (MUL_DRELU / MUL_DGELU),
    so hidden activations are read once and no elementwise pass exists.
    """

    @staticmethod
    def forward(ctx, x, residual, act, out_dtype, *params):
        n = len(params) // 2
        Kd = x.shape[-1]
        h = x.reshape(-1, Kd)
        if h.stride(1) != 1 or (h.stride(0) % 8) != 0:
            h = h.contiguous()
        inputs, pres = [h], []
        for i in range(n):
            w, b = params[2 * i], params[2 * i + 1]
            if i < n - 1:
                if act == "gelu":
                    pre = torch.empty((h.shape[0], w.shape[0]), dtype=K.adt(), device=h.device)
                    h = _fwd_gemm(h, w, b, None, epilogue=K.EPI_GELU, aux_out=pre)
                    pres.append(pre)
                else:
                    h = _fwd_gemm(h, w, b, None, epilogue=K.EPI_RELU)
                inputs.append(h)
            else:
                if residual is not None:
                    r2 = residual.reshape(-1, w.shape[0]).contiguous()
                    out = _fwd_gemm(h, w, b, None, epilogue=K.EPI_RESIDUAL, aux=r2, out_dtype=F32)
                else:
                    out = _fwd_gemm(h, w, b, None, out_dtype=out_dtype)
        ctx.save_for_backward(*inputs, *pres, *params)
        ctx.n, ctx.act, ctx.has_res, ctx.xshape, ctx.xdtype = n, act, residual is not None, x.shape, x.dtype
        return out.reshape(*x.shape[:-1], out.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        n, act = ctx.n, ctx.act
        sv = ctx.saved_tensors
        inputs = sv[:n]
        npre = n - 1 if act == "gelu" else 0
        pres = sv[n:n + npre]
        params = sv[n + npre:]
        g = dy.reshape(-1, dy.shape[-1])
        if g.stride(1) != 1 or (g.stride(0) % 8) != 0:
            g = g.contiguous()
        dx = None
        for i in reversed(range(n)):
            w, b = params[2 * i], params[2 * i + 1]
            _wgrad(g, inputs[i], w, b)
            if i > 0:
                if act == "gelu":
                    g = _dgrad_gemm(g, w, None, epilogue=K.EPI_MUL_DGELU, aux=pres[i - 1])
                else:
                    g = _dgrad_gemm(g, w, None, epilogue=K.EPI_MUL_DRELU, aux=inputs[i])
            elif ctx.needs_input_grad[0]:
                dx = _dgrad_gemm(g, w, None, out_dtype=ctx.xdtype).reshape(ctx.xshape)
        return (dx, (dy if ctx.has_res else None), None, None) + (None,) * len(params)


def mlp(x, params: Sequence[torch.Tensor], *, act="relu", residual=None, out_dtype=BF16):
    return MLPFn.apply(x, residual, act, out_dtype, *params)


# --------------------------------------------------------------------------------------------
# LayerNorm (optionally of x + res)
# --------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, out_dtype, eps):
        xc = x.contiguous()
        rc = res.contiguous() if res is not None else None
        need_h = rc is not None or xc.dtype != F32
        tw = [] if (out_dtype == F32 and xc.numel() // xc.shape[-1] >= _BIG_M) else None
        y, h, stats = K.layernorm_fwd(xc, gamma.detach(), beta.detach(), res=rc, out_dtype=out_dtype, want_h=need_h, eps=eps, twin=tw)
        if tw:
            _register_twin(y, tw[0])
        ctx.save_for_backward(h if need_h else xc, stats, gamma, beta)
        ctx.has_res, ctx.xdtype, ctx.rdtype = rc is not None, x.dtype, (res.dtype if res is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, stats, gamma, beta = ctx.saved_tensors
        dg = grad_buf(gamma) if gamma.requires_grad else None
        db = grad_buf(beta) if gamma.requires_grad else None
        tw = [] if h.numel() // h.shape[-1] >= _BIG_M else None
        dx = _ln_bwd(dy.contiguous(), h, stats, gamma.detach(), dgamma=dg, dbeta=db, twin=tw)
        if tw:
            _register_twin(dx, tw[0])
        _tn_notify(gamma, beta)          # (held until the parked dgamma / dbeta partials are summed)
        dxx = dx if ctx.xdtype == F32 else dx.to(ctx.xdtype)
        dr = None
        if ctx.has_res:
            dr = dx if ctx.rdtype == F32 else dx.to(ctx.rdtype)
        return dxx, dr, None, None, None, None


def layer_norm(x, gamma, beta, *, res=None, out_dtype=F32, eps=1e-5):
    """LN(x + res) over the last dim (HIP).  x / res fp32 or bf16; gradients flow to both."""
    return LayerNormFn.apply(x, res, gamma, beta, out_dtype, eps)


# --------------------------------------------------------------------------------------------
# Swin block: LN1 -> qkv -> fused window attention -> proj(+x) -> LN2 -> fc1+GELU -> fc2(+x)
# --------------------------------------------------------------------------------------------
def drop_path_scales(B: int, drop_prob: float):
    """timm DropPath (reference backbone/swin.py:8, 231, 279, 289) for one residual branch: per-sample multipliers
    floor(keep_prob + U[0,1)) / keep_prob as python floats.  Drawn from torch's CPU generator: no device sync."""
    keep = 1.0 - drop_prob
    return [float(v) / keep for v in torch.floor(keep + torch.rand(B)).tolist()]


_SCALE_VECS = {}


def _scale_vec(scales, device) -> torch.Tensor:
    """Device copy of a tuple of per-sample DropPath multipliers; cached (a block draws from {0, 1 / keep}^B: few distinct tuples)."""
    key = (tuple(scales), str(device))
    t = _SCALE_VECS.get(key)
    if t is None:
        if len(_SCALE_VECS) > 4096:
            _SCALE_VECS.clear()
        B, c = len(scales), max(max(scales), 0.0)
        if 0.0 < c and B <= 4:
            # a block's branch only ever draws from {0, c}^B: upload all combinations in ONE copy the first time c is seen (an upload
            # from pageable memory waits for the stream, so per-step misses would serialise host and GPU)
            combos = [[c if (m >> b) & 1 else 0.0 for b in range(B)] for m in range(1 << B)]
            allv = torch.tensor(combos, dtype=F32, device=device)
            for m, row in enumerate(combos):
                _SCALE_VECS[(tuple(row), str(device))] = allv[m]
            t = _SCALE_VECS.get(key)
        if t is None:
            t = torch.tensor(list(scales), dtype=F32, device=device)
            _SCALE_VECS[key] = t
    return t


def _branch_gemm(h, w16, bias, res, out_rows, scales):
    """out = res + scale_b * (h @ w^T + bias) per sample b (rows split evenly): the residual epilogue with alpha = scale_b, ONE
    GEMM over all images (uenc_gemm_nt_scaled); a dropped image's rows come out as the residual alone."""
    M = h.shape[0]
    if scales is None:
        return K.gemm_nt(h, w16, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=F32)
    B = len(scales)
    L = M // B
    if not K.EXACT:
        return K.gemm_nt(h, w16, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res, out_dtype=F32, sample_scale=_scale_vec(scales, h.device),
                         rows_per_sample=L)
    out = torch.empty((M, out_rows), dtype=F32, device=h.device)       # fp32 verification mode: image by image
    for b, sc in enumerate(scales):
        rows = slice(b * L, (b + 1) * L)
        if sc == 0.0:
            out[rows].copy_(res[rows])                 # dropped branch: the residual passes through
        else:
            K.gemm_nt(h[rows], w16, bias=bias, epilogue=K.EPI_RESIDUAL, aux=res[rows], out=out[rows], alpha=sc)
    return out


def _scaled_rows(g16, scales):
    """bf16 copy of a (M, C) gradient with sample b's rows multiplied by scale_b (the backward of the DropPath scaling).
    Only the fp32 verification mode takes this path; the bf16 mode scales inside the GEMM epilogues (`_branch_dgrad`, `_branch_wgrad`)."""
    if scales is None:
        return g16
    B = len(scales)
    sc = torch.tensor(scales, dtype=torch.float32).to(g16.device, non_blocking=True)
    return (g16.view(B, -1, g16.shape[-1]).float() * sc.view(B, 1, 1)).to(K.adt()).view(g16.shape)


def _branch_dgrad(g16, w16t, scales, **kw):
    """(scale_b * g) @ w for a DropPath-scaled branch: the per-sample scale rides in the dgrad GEMM's epilogue (it commutes with the
    contraction and with the activation-derivative product of `kw`)."""
    if scales is None or K.EXACT:
        return K.gemm_nt(g16, w16t, **kw)
    return K.gemm_nt(g16, w16t, sample_scale=_scale_vec(scales, g16.device), rows_per_sample=g16.shape[0] // len(scales), **kw)


def _branch_wgrad(g16, x, gw, gb, notify, scales):
    """gw += (scale * g)^T x, gb += column sums of scale * g, for a DropPath-scaled branch and the UNSCALED gradient g16: the kept
    images share one multiplier 1 / keep, so the sum runs over their rows only and is scaled once (alpha of the TN GEMM).  Exactly
    one gradient-ready notification whatever was dropped (the data-parallel bucket scheduler counts contributions)."""
    if scales is None or K.EXACT:
        _tn(g16, x, gw, gb, notify)
        return
    B = len(scales)
    L = g16.shape[0] // B
    runs, b = [], 0
    while b < B:                                       # maximal runs of kept images = contiguous row ranges
        if scales[b] != 0.0:
            e = b
            while e + 1 < B and scales[e + 1] != 0.0:
                e += 1
            runs.append((b, e + 1))
            b = e + 1
        else:
            b += 1
    if not runs:
        if WGRADS.busy():
            WGRADS.notify.extend(p for p in notify if p is not None)
        else:
            _notify(*notify)
        return
    alpha = float(max(scales))
    for i, (b0, b1) in enumerate(runs):
        rows = slice(b0 * L, b1 * L)
        _tn(g16[rows], x[rows], gw, gb, notify if i == len(runs) - 1 else (), alpha=alpha)


class SwinBlockFn(torch.autograd.Function):
    """One whole SwinTransformerBlock (reference backbone/swin.py:235-295) as 7 kernels forward and
    13 backward, all HIP.  x is the fp32 residual stream (B, L, C).  dp: None (eval) or (scales_attn, scales_mlp), the
    per-sample DropPath multipliers of the two residual branches in training mode (`drop_path_scales`)."""

    @staticmethod
    def forward(ctx, x, H, W, ws, shift, nH, scale, dp, g1, b1, wqkv, bqkv, table, wproj, bproj, g2, b2, w1, bb1, w2, bb2):
        B, L, C = x.shape
        s1, s2 = dp if dp is not None else (None, None)
        M = B * L
        x2 = x.reshape(M, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        xn, _, st1 = K.layernorm_fwd(x2, g1.detach(), b1.detach(), out_dtype=BF16)
        qkv = K.gemm_nt(xn, CACHE.mat(wqkv), bias=bqkv.detach())
        bias_q = bias_k = CACHE.relpos(table, ws)         # (the kernels read bias_q only; refreshed for all blocks by one launch per step)
        # the softmax row statistics are kept for the backward of 12 x 12 windows (its kernel then skips the maximum / sum passes)
        want_lse = ws == 12 and not K.EXACT and any(ctx.needs_input_grad)
        attn = K.window_attn_fwd(qkv.view(B, H, W, 3 * C), CACHE.vec16(bqkv), bias_q, ws, shift, scale, want_lse=want_lse)
        attn, lse = attn if want_lse else (attn, None)
        ctx.has_lse = lse is not None
        x1 = _branch_gemm(attn.view(M, C), CACHE.mat(wproj), bproj.detach(), x2, C, s1)
        xn2, _, st2 = K.layernorm_fwd(x1, g2.detach(), b2.detach(), out_dtype=BF16)
        pre = torch.empty((M, w1.shape[0]), dtype=K.adt(), device=x.device)
        h = K.gemm_nt(xn2, CACHE.mat(w1), bias=bb1.detach(), epilogue=K.EPI_GELU, aux_out=pre)
        x2o = _branch_gemm(h, CACHE.mat(w2), bb2.detach(), x1, C, s2)
        ctx.dp = (s1, s2)
        ctx.save_for_backward(x2, st1, xn, qkv, bias_q, bias_k, attn, x1, st2, xn2, pre, h,
                              g1, b1, wqkv, bqkv, table, wproj, bproj, g2, b2, w1, bb1, w2, bb2, *([lse] if lse is not None else []))
        ctx.geom = (B, L, C, H, W, ws, shift, nH, scale)
        return x2o.view(B, L, C)

    @staticmethod
    def backward(ctx, dxo):
        (x2, st1, xn, qkv, bias_q, bias_k, attn, x1, st2, xn2, pre, h,
         g1, b1, wqkv, bqkv, table, wproj, bproj, g2, b2, w1, bb1, w2, bb2) = ctx.saved_tensors[:25]
        lse = ctx.saved_tensors[25] if ctx.has_lse else None
        B, L, C, H, W, ws, shift, nH, scale = ctx.geom
        M = B * L
        d2 = dxo.reshape(M, C)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        train = wqkv.requires_grad
        # MLP branch (bf16 copy of the incoming stream gradient: the dgrad / wgrad GEMMs read bf16 operands)
        s1, s2 = ctx.dp
        d2h = _twin(d2)                                     # written by the LayerNorm backward that produced this gradient
        if d2h is None:
            d2h = K.cast_bf16(d2)
        if K.EXACT:
            d2h, s2 = _scaled_rows(d2h, s2), None           # (verification mode: the scaled gradient is materialised)
        # DropPath: the MLP branch sees scale_b * gradient -- applied in the dgrad epilogue and as the wgrad's alpha
        dh = _branch_dgrad(d2h, CACHE.mat_t(w2), s2, epilogue=K.EPI_MUL_DGELU, aux=pre)     # (M, 4C) d(pre-GELU)
        if train:
            _branch_wgrad(d2h, h, grad_buf(w2), grad_buf(bb2), (w2, bb2), s2)
        dxn2 = K.gemm_nt(dh, CACHE.mat_t(w1))                                              # (M, C)
        if train:
            _tn(dh, xn2, grad_buf(w1), grad_buf(bb1), (w1, bb1))
        tw = []
        dx1 = _ln_bwd(dxn2, x1, st2, g2.detach(), dres=d2,
                              dgamma=grad_buf(g2) if train else None, dbeta=grad_buf(b2) if train else None, twin=tw)
        # attention branch
        dx1h = tw[0]
        if K.EXACT:
            dx1h, s1 = _scaled_rows(dx1h, s1), None
        dattn = _branch_dgrad(dx1h, CACHE.mat_t(wproj), s1)                                # (M, C) bf16
        if train:
            _branch_wgrad(dx1h, attn.view(M, C), grad_buf(wproj), grad_buf(bproj), (wproj, bproj), s1)
        # (the kernels add the relative-position-table gradient and the padding-slot share of the qkv-bias gradient straight into .grad)
        dqkv = _wattn_bwd(qkv.view(B, H, W, 3 * C), CACHE.vec16(bqkv), bias_q, bias_k, attn, dattn.view(B, H, W, C), ws, shift, scale,
                                 dtable=grad_buf(table) if train else None, dbias=grad_buf(bqkv) if train else None, lse=lse)
        dqkv2 = dqkv.view(M, 3 * C)
        dxn = K.gemm_nt(dqkv2, CACHE.mat_t(wqkv))
        if train:
            _tn(dqkv2, xn, grad_buf(wqkv), grad_buf(bqkv), (wqkv, bqkv))
        tw = []
        dx = _ln_bwd(dxn, x2, st1, g1.detach(), dres=dx1,
                             dgamma=grad_buf(g1) if train else None, dbeta=grad_buf(b1) if train else None, twin=tw)
        _register_twin(dx, tw[0])                           # the previous block's backward starts from dx in bf16
        if train:
            _tn_notify(g1, b1, table, g2, b2)
        return (dx.view(B, L, C),) + (None,) * 20


def swin_block(x, H, W, ws, shift, nH, scale, params: Sequence[torch.Tensor], dp=None):
    return SwinBlockFn.apply(x, H, W, ws, shift, nH, scale, dp, *params)


# --------------------------------------------------------------------------------------------
# One whole deformable-attention encoder layer (reference pixel_decoder/msdeformattn.py:103-142 with
# ops/modules/ms_deform_attn.py:74-113) as one autograd node: 10 kernels forward, explicit backward.
# --------------------------------------------------------------------------------------------
class DeformEncoderLayerFn(torch.autograd.Function):
    """src (B, S, C) fp32 stream -> LN2(src1 + FFN(src1)),  src1 = LN1(src + MSDeformAttn(src + pos, src)).

    What the module-by-module form spends on glue disappears here: query = bf16(src + pos) is one kernel, sampling offsets
    and attention logits come from ONE GEMM over the stacked weights, softmax + sampling-location arithmetic is one small
    kernel each way, every residual / skip-path sum rides in a GEMM or LayerNorm epilogue (no elementwise adds), the
    LayerNorms hand their results on in fp32 and bf16 at once (no casts), and the gradient of the level embedding (which
    enters through pos) is taken from per-level column sums of the offset/logit gradient instead of a (B, S, C) map.
    """

    @staticmethod
    def forward(ctx, src, pos, level_embed, ref, shapes, level_start, nH, nP, drop,
                wv, bv, woff, boff, waw, baw, wo, bo, g1, b1, w1, bb1, w2, bb2, g2, b2):
        B, S, C = src.shape
        M, L, D = B * S, shapes.shape[0], C // nH
        x = src.reshape(M, C)
        x = x if x.is_contiguous() else x.contiguous()
        x16 = _twin(x)
        if x16 is None:
            x16 = K.cast_bf16(x)
        posc = pos.detach()
        posc = posc if posc.is_contiguous() else posc.contiguous()
        q16 = K.add_cast_bf16(x, posc.reshape(-1, C))
        value = K.gemm_nt(x16, CACHE.mat(wv), bias=bv.detach())                                   # (M, C) bf16
        offaw = K.gemm_nt(q16, CACHE.cat(woff, waw), bias=CACHE.catvec(boff, baw), out_dtype=F32)   # (M, 3 nH L P) fp32
        shapes_host = _host_shapes(shapes)
        # fused: sampling locations / softmaxed weights are derived inside the attention kernels from the projection row (no loc / aw tensors,
        # no glue kernels); UENC_MSDA_FUSED=0 keeps the module-by-module pair (A/B, and the form every other configuration takes)
        fused = os.environ.get("UENC_MSDA_FUSED", "1") != "0" and K.msdeform_fused_available(shapes_host, B, nH, D, L, S, nP)
        # (the general gather kernel keeps the glue kernel + core pair in the forward -- its fused form measured 5 % slower, tools/msda_fused_bench.py;
        # the backward recomputes locations / weights from the saved projection row)
        v4 = value.view(B, S, nH, D)
        if (fused and nP == 4 and offaw.stride(0) % 4 == 0 and K.msdeform_tiled_eligible(v4, shapes_host, S, L, nP)
                and os.environ.get("UENC_MSDA_FUSED_FWD", "1") != "0"):
            # the queries are the maps' own pixels: the LDS-tiled kernel derives locations / weights itself (no glue kernel, no loc / aw tensors)
            att = K.msdeform_attn_fused_fwd(v4, shapes, level_start, offaw, ref, L, nP, out_dtype=BF16, shapes_host=shapes_host).view(M, C)
            loc, aw = offaw, ref
        else:
            loc, aw = K.msda_prep_fwd(offaw, ref, shapes, B, S, nH, L, nP)
            att = K.msdeform_attn_fwd(v4, shapes, level_start, loc, aw, out_dtype=BF16, shapes_host=shapes_host).view(M, C)
            if fused:
                loc, aw = offaw, ref
        if drop is None or K.EXACT:
            assert drop is None, "the fp32 verification mode has no dropout path"
            # both post-norms ride in the epilogue of the residual GEMM in front of them (d_model fits one column tile): no LayerNorm pass
            fz = K.gemm_nt_ln(att, CACHE.mat(wo), bo.detach(), x, g1.detach(), b1.detach())
            if fz is not None:
                h1, s1, s1_16, st1 = fz
            else:
                h1 = K.gemm_nt(att, CACHE.mat(wo), bias=bo.detach(), epilogue=K.EPI_RESIDUAL, aux=x, out_dtype=F32)
                tw = []
                s1, _, st1 = K.layernorm_fwd(h1, g1.detach(), b1.detach(), out_dtype=F32, twin=tw)
                s1_16 = tw[0]
            f = K.gemm_nt(s1_16, CACHE.mat(w1), bias=bb1.detach(), epilogue=K.EPI_RELU)                # (M, ffn) bf16
            fz = K.gemm_nt_ln(f, CACHE.mat(w2), bb2.detach(), s1, g2.detach(), b2.detach())
            if fz is not None:
                h2, out, o16, st2 = fz
                tw = [o16]
            else:
                h2 = K.gemm_nt(f, CACHE.mat(w2), bias=bb2.detach(), epilogue=K.EPI_RESIDUAL, aux=s1, out_dtype=F32)
                tw = []
                out, _, st2 = K.layernorm_fwd(h2, g2.detach(), b2.detach(), out_dtype=F32, twin=tw)
        else:
            # training: dropout1 / 2 / 3 of the reference layer (:111-142) between the same kernels.  The branch outputs leave their GEMMs
            # as bf16, are masked in place (index-hash keep mask: nothing stored), and the residual sums move into the LayerNorm kernels.
            pd, sd1, sd2, sd3 = drop
            hb = K.dropout_bf16(K.gemm_nt(att, CACHE.mat(wo), bias=bo.detach()), sd1, pd)
            tw = []
            s1, h1, st1 = K.layernorm_fwd(hb, g1.detach(), b1.detach(), res=x, out_dtype=F32, want_h=True, twin=tw)
            s1_16 = tw[0]
            f = K.gemm_nt(s1_16, CACHE.mat(w1), bias=bb1.detach(), epilogue=K.EPI_RELU)
            K.dropout_bf16(f, sd2, pd, out=f)
            hb = K.dropout_bf16(K.gemm_nt(f, CACHE.mat(w2), bias=bb2.detach()), sd3, pd)
            tw = []
            out, h2, st2 = K.layernorm_fwd(hb, g2.detach(), b2.detach(), res=s1, out_dtype=F32, want_h=True, twin=tw)
            del hb
        _register_twin(out, tw[0])                                                                 # the next layer's value / query operand source
        ctx.save_for_backward(x16, q16, value, loc, aw, att, h1, st1, s1_16, f, h2, st2, shapes, level_start,
                              wv, bv, woff, boff, waw, baw, wo, bo, g1, b1, w1, bb1, w2, bb2, g2, b2)
        ctx.geom = (B, S, C, nH, nP)
        ctx.drop = drop
        ctx.shapes_host = shapes_host
        ctx.fused = fused               # then (loc, aw) hold (offaw, ref)
        return out.view(B, S, C)

    @staticmethod
    def backward(ctx, dout):
        (x16, q16, value, loc, aw, att, h1, st1, s1_16, f, h2, st2, shapes, level_start,
         wv, bv, woff, boff, waw, baw, wo, bo, g1, b1, w1, bb1, w2, bb2, g2, b2) = ctx.saved_tensors
        B, S, C, nH, nP = ctx.geom
        M, L, D = B * S, shapes.shape[0], C // nH
        train = wv.requires_grad
        dy = dout.reshape(M, C)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        gb = (lambda p: grad_buf(p)) if train else (lambda p: None)
        # FFN block
        tw = []
        dh2 = _ln_bwd(dy, h2, st2, g2.detach(), dgamma=gb(g2), dbeta=gb(b2), twin=tw)       # also the skip-path gradient of s1
        dh2_16 = tw[0]
        drop = ctx.drop
        inv_keep = 1.0
        if drop is not None:            # the branch gradients are the masked, rescaled stream gradients; f > 0 <=> ReLU active AND kept
            K.dropout_bf16(dh2_16, drop[3], drop[0], out=dh2_16)
            inv_keep = 1.0 / (1.0 - drop[0])
        df = K.gemm_nt(dh2_16, CACHE.mat_t(w2), epilogue=K.EPI_MUL_DRELU, aux=f, alpha=inv_keep)    # (M, ffn) bf16
        ds1 = K.gemm_nt(df, CACHE.mat_t(w1), epilogue=K.EPI_RESIDUAL, aux=dh2, out_dtype=F32)      # dh2 + dFFN-in
        if train:
            _tn(dh2_16, f, grad_buf(w2), grad_buf(bb2), (w2, bb2))
            _tn(df, s1_16, grad_buf(w1), grad_buf(bb1), (w1, bb1))
        # attention block
        tw = []
        dh1 = _ln_bwd(ds1, h1, st1, g1.detach(), dgamma=gb(g1), dbeta=gb(b1), twin=tw)      # also the skip-path gradient of src
        dh1_16 = tw[0]
        if drop is not None:
            K.dropout_bf16(dh1_16, drop[1], drop[0], out=dh1_16)
        datt = K.gemm_nt(dh1_16, CACHE.mat_t(wo))                                                  # (M, C) bf16
        if train:
            _tn(dh1_16, att, grad_buf(wo), grad_buf(bo), (wo, bo))
        ncol = 3 * nH * L * nP
        if ctx.fused:
            gv, doffaw = K.msdeform_attn_fused_bwd(value.view(B, S, nH, D), shapes, level_start, loc, aw, L, nP, datt.view(B, S, C), ctx.shapes_host)
        else:
            gv, gl, ga = K.msdeform_attn_bwd(value.view(B, S, nH, D), shapes, level_start, loc, aw, datt.view(B, S, C), ctx.shapes_host)
            doffaw = K.msda_prep_bwd(gl, ga, aw, shapes, ncol)                                     # (M, ncol) bf16
        gv16 = K.cast_bf16(gv.view(M, C))
        dsrc = K.gemm_nt(doffaw, CACHE.cat(woff, waw, transposed=True), epilogue=K.EPI_RESIDUAL, aux=dh1, out_dtype=F32)
        dsrc = K.gemm_nt(gv16, CACHE.mat_t(wv), epilogue=K.EPI_RESIDUAL, aux=dsrc, out=dsrc)        # dh1 + dq + dvalue-in
        dlev = None
        if train:
            no = 2 * nH * L * nP
            _tn(doffaw[:, :no], q16, grad_buf(woff), grad_buf(boff), (woff, boff))
            _tn(doffaw[:, no:], q16, grad_buf(waw), grad_buf(baw), (waw, baw))
            _tn(gv16, x16, grad_buf(wv), grad_buf(bv), (wv, bv))
            _tn_notify(g1, b1, g2, b2)
        if ctx.needs_input_grad[2]:
            # d level_embed[l] = sum over the level's tokens of dq = (per-level column sums of doffaw) @ [Woff; Waw]
            sums = K.segment_colsum(doffaw, level_start, S, B)                                      # (L, ncol) fp32
            wcat = torch.cat([woff.detach().reshape(woff.shape[0], -1), waw.detach().reshape(waw.shape[0], -1)], 0)
            dlev = sums @ wcat
        return (dsrc.view(B, S, C), None, dlev, None, None, None, None, None, None) + (None,) * 16


def deform_encoder_layer(src, pos, level_embed, ref, shapes, level_start, nH, nP, params: Sequence[torch.Tensor], drop=None):
    """drop: None, or (p, seed1, seed2, seed3) for the layer's three dropouts in training mode."""
    return DeformEncoderLayerFn.apply(src, pos, level_embed, ref, shapes, level_start, nH, nP, drop, *params)


# --------------------------------------------------------------------------------------------
# multi-scale deformable attention core (the reference's MSDeformAttnFunction, same argument order)
# --------------------------------------------------------------------------------------------
_HOST_SHAPES = {}


def _host_shapes(shapes: torch.Tensor):
    """Host copy of a (L, 2) spatial-shapes tensor; one device sync per distinct tensor object (the pixel decoder keeps
    its geometry tensors alive across steps), none afterwards."""
    ent = _HOST_SHAPES.get(id(shapes))
    if ent is not None and ent[0]() is shapes and ent[1] == shapes._version:
        return ent[2]
    if len(_HOST_SHAPES) > 64:
        _HOST_SHAPES.clear()
    host = [tuple(int(v) for v in hw) for hw in shapes.tolist()]
    _HOST_SHAPES[id(shapes)] = (weakref.ref(shapes), shapes._version, host)
    return host


class MSDeformAttnFunction(torch.autograd.Function):
    """pixel_decoder/ops/functions/ms_deform_attn_func.py:35-52: forward(value, spatial_shapes,
    level_start_index, sampling_locations, attention_weights, im2col_step) -> (N, Lq, M*D)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=128, out_dtype=None):
        for t in (value, sampling_locations, attention_weights):
            if not t.is_cuda:
                raise RuntimeError("Not implemented on the CPU")        # ms_deform_attn.h:43
            if not t.is_contiguous():
                raise RuntimeError("tensor has to be contiguous")        # ms_deform_attn_cuda.cu:33-37
        ctx.shapes_host = _host_shapes(value_spatial_shapes)        # lets the backward bin grad_value per block of pixels (and the forward tile)
        out = K.msdeform_attn_fwd(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                  attention_weights, out_dtype=out_dtype or (F32 if value.dtype == F32 else BF16), shapes_host=ctx.shapes_host)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        gv, gl, ga = K.msdeform_attn_bwd(value, shapes, start, loc, attn, grad_output.contiguous(), ctx.shapes_host)
        return (gv if value.dtype == F32 else gv.to(value.dtype)), None, None, gl, ga, None, None


# --------------------------------------------------------------------------------------------
# mask einsum  "bqc,bchw->bqhw"  as batched NT GEMMs against channels-last mask features
# --------------------------------------------------------------------------------------------
def mask_logits_eager(me, mf16_tok):
    """me (B, Q, C) bf16 x mask features (B, HW, C) bf16 -> (B, Q, HW) fp32 mask logits, outside autograd (see MaskHeadsFn)."""
    B, Q, C = me.shape
    HW = mf16_tok.shape[1]
    assert HW % 8 == 0 and C % 8 == 0
    out = torch.empty((B, Q, HW), dtype=F32, device=me.device)
    med = me.detach()
    for b in range(B):
        K.gemm_nt(med[b], mf16_tok[b], out=out[b])
    return out


class MaskHeadsFn(torch.autograd.Function):
    """All prediction heads' mask einsums "bqc,bchw->bqhw" as ONE autograd node.

    The decoder needs every head's mask logits during its forward (they become the next layer's attention mask, a detached
    threshold), so they are computed eagerly (`mask_logits_eager`) and this node only adopts them as its outputs.  What it buys
    is the backward: the loss gradients of all heads are available together, so
      * d(mask embeddings) of all heads is one 1536 x 256 x HW GEMM per image (stacked bf16 gradient rows x mask features)
        instead of a 150-row, 64-way split-K GEMM per head and image,
      * d(mask features) is one token-contraction GEMM per image, stored once (no atomics, no accumulation passes over the
        134 MB map per head).
    """

    @staticmethod
    def forward(ctx, mf32_tok, mf16_chw, pre, *mes):
        ctx.save_for_backward(mf16_chw, *mes)
        ctx.shape = tuple(mf32_tok.shape)
        return tuple(pre)

    @staticmethod
    def backward(ctx, *douts):
        mf16_chw, *mes = ctx.saved_tensors
        B, HW, C = ctx.shape
        n, Q = len(mes), mes[0].shape[1]
        Mp = -(-(n * Q) // 64) * 64
        dev = mf16_chw.device
        D = torch.empty((B, Mp, HW), dtype=K.adt(), device=dev)       # stacked gradient rows of all heads
        ME = torch.zeros((B, Mp, C), dtype=K.adt(), device=dev)
        for h, (d, me) in enumerate(zip(douts, mes)):
            r = h * Q
            if d is None:
                D[:, r:r + Q].zero_()
                continue
            d = d if d.is_contiguous() else d.contiguous()
            for b in range(B):
                K.cast_bf16(d[b], out=D[b, r:r + Q])
            ME[:, r:r + Q] = me
        if n * Q < Mp:
            D[:, n * Q:].zero_()                                      # (0 x garbage could be NaN)
        # d(mask embeddings): rows = heads x queries, contraction over the HW pixels
        # (split over the pixels so that every CU has a tile; partial tiles are stored and summed, not added atomically)
        tiles = -(-Mp // 256) * -(-C // 256)
        split = max(1, min(HW // 512, -(-256 // tiles)))
        dme = [K.gemm_nt_splitk(D[b], mf16_chw[b], split) for b in range(B)]
        # d(mask features): contraction over the stacked rows, one stored GEMM per image
        dmf = None
        if ctx.needs_input_grad[0] and K.EXACT:
            dmf = torch.zeros((B, HW, C), dtype=F32, device=dev)
            for b in range(B):
                K.gemm_tn(D[b], ME[b], dmf[b], None)
        elif ctx.needs_input_grad[0]:
            dmf = torch.empty((B, HW, C), dtype=F32, device=dev)
            tile = 256 if C % 256 == 0 else 128
            tiles_k = -(-C // tile)
            items = -(-HW // tile) * tiles_k
            WgradQueue.launch(tile, [(D[b].data_ptr(), ME[b].data_ptr(), dmf[b].data_ptr(), 0, D.stride(1), C, C, Mp, HW, C, tiles_k, Mp, 1, items, 1)
                                     for b in range(B)], dev)
        return (dmf, None, None) + tuple(torch.stack([dme[b][h * Q:(h + 1) * Q] for b in range(B)]).to(mes[h].dtype) for h in range(n))


def mask_heads(mf32_tok, mf16_chw, pre, mes):
    """pre: the eager mask logits per head (list of (B, Q, HW) fp32); mes: the heads' mask embeddings (B, Q, C) bf16."""
    return MaskHeadsFn.apply(mf32_tok, mf16_chw, list(pre), *mes)


# --------------------------------------------------------------------------------------------
# multi-head attention core of the decoder (head_dim 32): q (B, Lq, E), k / v (B, S, E)
# --------------------------------------------------------------------------------------------
def attention(q, k, v, nheads: int, mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """softmax(q k^T / sqrt(d) [blocked where mask]) v, heads interleaved in E.  -> (B, Lq, E) bf16.
    dropout_p > 0: dropout on the attention probabilities (training), keep-mask derived from `seed` inside the kernels."""
    from .attention import mha
    return mha(q, k, v, nheads, mask, dropout_p, seed)


# --------------------------------------------------------------------------------------------
# 3x3 convolution (stride 1, pad 1, no bias) on channels-last maps as im2col + MFMA GEMM
# --------------------------------------------------------------------------------------------
class GroupNormTokensFn(torch.autograd.Function):
    """torch.nn.GroupNorm on a token matrix (B, HW, C) [+ bilinear top-down merge] [+ ReLU], channels-last throughout
    (reference pixel_decoder/msdeformattn.py:283-302 norm layers, :343-352 merge)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, relu, add_src, add_hw, out_dtype, dx_dtype):
        xc = x if x.is_contiguous() else x.contiguous()
        src = None
        if add_src is not None:
            src = add_src.float()
            src = src if src.is_contiguous() else src.contiguous()
        y, stats = K.groupnorm_tokens_fwd(xc, gamma.detach(), beta.detach(), G, eps, relu=relu, add_src=src, add_hw=add_hw,
                                          out_dtype=out_dtype)
        ctx.save_for_backward(xc, gamma, beta, stats)
        ctx.cfg = (G, relu, add_hw, None if add_src is None else tuple(add_src.shape), dx_dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        G, relu, add_hw, src_shape, dx_dtype = ctx.cfg
        dy = dy if dy.is_contiguous() else dy.contiguous()
        train = gamma.requires_grad
        dx = K.groupnorm_tokens_bwd(dy, x, gamma.detach(), beta.detach(), stats, G, relu=relu,
                                    dgamma=grad_buf(gamma) if train else None, dbeta=grad_buf(beta) if train else None,
                                    dx_dtype=dx_dtype)
        if train:
            _tn_notify(gamma, beta)
        dsrc = None
        if src_shape is not None and ctx.needs_input_grad[6]:
            B, Hs, Ws, C = src_shape
            dsrc = K.upsample_bilinear_tokens_bwd(dy.view(B, add_hw[0], add_hw[1], C), Hs, Ws)      # (the ReLU is never combined with a merge)
        return dx, None, None, None, None, None, dsrc, None, None, None


def group_norm_tokens(x, gn, *, relu=False, add_src=None, add_hw=None, out_dtype=F32, dx_dtype=BF16):
    """`gn`: an nn.GroupNorm; x (B, HW, C); add_src (B, Hs, Ws, C) fp32 is resized to add_hw = (H, W) and added."""
    assert not (relu and add_src is not None)
    return GroupNormTokensFn.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, add_src, add_hw, out_dtype, dx_dtype)


class Conv3x3Fn(torch.autograd.Function):
    """The single 3x3 conv of the FPN (reference pixel_decoder/msdeformattn.py:293-302 `layer_1`, 154 GFLOP per image
    at 1/4 resolution).  x (B, H, W, Cin) bf16 channels-last; weight (Cout, Cin, 3, 3) -> (B, H*W, Cout) fp32.  Forward:
    patch matrix (B*H*W, 9*Cin) + the 256x256 LDS-DMA GEMM; backward: dgrad GEMM + col2im gather, wgrad through the
    token-contraction GEMM on the saved patch matrix."""

    @staticmethod
    def _wmat(weight):
        return CACHE._get(weight, "c3", lambda: K.cast_bf16(weight.detach().permute(0, 2, 3, 1).reshape(weight.shape[0], -1).contiguous()))

    @staticmethod
    def _wmat_t(weight):
        return CACHE._get(weight, "c3t", lambda: K.cast_transpose_bf16(
            weight.detach().permute(0, 2, 3, 1).reshape(weight.shape[0], -1).contiguous()))

    @staticmethod
    def forward(ctx, x, weight):
        B, H, W, C = x.shape
        x16 = x if x.dtype == K.adt() else x.to(K.adt())
        col = K.im2col3x3(x16 if x16.is_contiguous() else x16.contiguous())
        out = K.gemm_nt(col, Conv3x3Fn._wmat(weight), out_dtype=F32)
        ctx.save_for_backward(col, weight)
        ctx.shape = (B, H, W, C)
        ctx.in_dtype = x.dtype
        return out.view(B, H * W, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        col, weight = ctx.saved_tensors
        B, H, W, C = ctx.shape
        Co = weight.shape[0]
        dy2 = dy.reshape(B * H * W, Co)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        if dy2.dtype != BF16:
            dy2 = K.cast_bf16(dy2.float())
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = K.gemm_nt(dy2, Conv3x3Fn._wmat_t(weight))                                  # (M, 9*Cin) bf16
            dx = K.col2im3x3(dcol, B, H, W, C)
            if ctx.in_dtype != dx.dtype:
                dx = dx.to(ctx.in_dtype)
        if weight.requires_grad:
            M = B * H * W
            dw = torch.zeros((Co, 9 * C), dtype=F32, device=dy.device)
            if M % 64 == 0 and Co % 8 == 0 and not K.EXACT:
                tile = 256 if Co % 256 == 0 else 128
                nsplit = max(1, M // 16384)
                mlen = -(-(M // 64) // nsplit) * 64
                nsplit = -(-M // mlen)
                tiles_k = -(-(9 * C) // tile)
                WgradQueue.launch(tile, [(dy2.data_ptr(), col.data_ptr(), dw.data_ptr(), 0, Co, 9 * C, 9 * C, M, Co, 9 * C, tiles_k, mlen,
                                          nsplit, -(-Co // tile) * tiles_k * nsplit, 0)], dy.device)
            else:
                K.gemm_tn(dy2, col, dw, None)
            grad_buf(weight).add_(dw.view(Co, 3, 3, C).permute(0, 3, 1, 2))
            _tn_notify(weight)
        return dx, None


def conv3x3(x_nhwc, weight):
    """x (B, H, W, Cin) bf16|fp32 channels-last -> (B, H*W, Cout) fp32."""
    return Conv3x3Fn.apply(x_nhwc, weight)


class _SubCtx:
    """The part of an autograd ctx the Functions above use, for running one Function inside another (below)."""

    def __init__(self, needs):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class LinearGroupNormFn(torch.autograd.Function):
    """1x1 conv (Linear, fp32 result) -> GroupNorm (+ bilinear top-down merge | + ReLU) on token matrices as ONE autograd node
    (reference pixel_decoder/msdeformattn.py:283-302, 343-352: Conv2d with norm=GroupNorm).  The arithmetic is LinearFn's and
    GroupNormTokensFn's, called in sequence; what the fusion buys is in the backward: the GroupNorm's input gradient stays bf16
    between the two -- as two nodes the engine cast it to fp32 (the conv output's dtype) and the Linear's backward cast it straight
    back for its GEMMs: two passes over the (B, HW, C) map per GroupNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, G, eps, relu, add_src, add_hw, out_dtype):
        ca = _SubCtx((ctx.needs_input_grad[0], False, False, False, False, False))
        y = LinearFn.forward(ca, x, weight, bias, None, None, F32)
        cb = _SubCtx((True, False, False, False, False, False, ctx.needs_input_grad[8], False, False, False))
        out = GroupNormTokensFn.forward(cb, y, gamma, beta, G, eps, relu, add_src, add_hw, out_dtype, BF16)
        ctx.sub = (ca, cb)
        return out

    @staticmethod
    def backward(ctx, dy):
        ca, cb = ctx.sub
        ctx.sub = None
        g = GroupNormTokensFn.backward(cb, dy)             # g[0]: d(conv output) bf16, g[6]: d(add_src)
        dx = LinearFn.backward(ca, g[0])[0]
        return dx, None, None, None, None, None, None, None, g[6], None, None


class Conv3x3GroupNormFn(torch.autograd.Function):
    """3x3 conv (im2col + GEMM, fp32 result) -> GroupNorm (+ ReLU) as one autograd node: see LinearGroupNormFn."""

    @staticmethod
    def forward(ctx, x_nhwc, weight, gamma, beta, G, eps, relu, out_dtype):
        ca = _SubCtx((ctx.needs_input_grad[0], False))
        y = Conv3x3Fn.forward(ca, x_nhwc, weight)
        cb = _SubCtx((True, False, False, False, False, False, False, False, False, False))
        out = GroupNormTokensFn.forward(cb, y, gamma, beta, G, eps, relu, None, None, out_dtype, BF16)
        ctx.sub = (ca, cb)
        return out

    @staticmethod
    def backward(ctx, dy):
        ca, cb = ctx.sub
        ctx.sub = None
        g = GroupNormTokensFn.backward(cb, dy)
        dx = Conv3x3Fn.backward(ca, g[0])[0]
        return dx, None, None, None, None, None, None, None


def linear_group_norm(x, conv, gn, *, relu=False, add_src=None, add_hw=None, out_dtype=F32):
    """conv: a 1x1 Conv2d / Linear (weight, bias); gn: nn.GroupNorm.  -> (B, HW, C)."""
    return LinearGroupNormFn.apply(x, conv.weight, conv.bias, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, add_src, add_hw, out_dtype)


def conv3x3_group_norm(x_nhwc, weight, gn, *, relu=False, out_dtype=F32):
    return Conv3x3GroupNormFn.apply(x_nhwc, weight, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, out_dtype)


# --------------------------------------------------------------------------------------------
# DiNAT: neighbourhood attention, the whole NATLayer, the 3x3 stride-2 convolutions
# --------------------------------------------------------------------------------------------
class NA2DFn(torch.autograd.Function):
    """natten2dqkrpb + softmax + natten2dav of natten.NeighborhoodAttention2D (reference call site backbone/dinat.py:77-79) on the
    qkv Linear's output (B, H, W, 3C) bf16 -> (B, H, W, C) bf16.  H, W >= ks * dilation (the module pads first, like NATTEN)."""

    @staticmethod
    def forward(ctx, qkv, rpb, nH, ks, dilation, scale):
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        rp = None if rpb is None else rpb.detach().float().contiguous()
        out, lse = K.na2d_fwd(qkv, rp, nH, ks, dilation, scale)
        ctx.save_for_backward(qkv, rpb, out, lse)
        ctx.cfg = (nH, ks, dilation, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, rpb, out, lse = ctx.saved_tensors
        nH, ks, dilation, scale = ctx.cfg
        dout = dout if dout.dtype == BF16 else dout.to(BF16)
        train = rpb is not None and rpb.requires_grad
        dqkv = K.na2d_bwd(qkv, None if rpb is None else rpb.detach().float().contiguous(), out, dout.contiguous(), lse, nH, ks, dilation, scale,
                          grad_buf(rpb) if train else None)
        if train:
            _notify(rpb)
        return dqkv, None, None, None, None, None


def na2d(qkv, rpb, nH: int, ks: int, dilation: int, scale: float):
    return NA2DFn.apply(qkv, rpb, nH, ks, dilation, scale)


class NATLayerFn(torch.autograd.Function):
    """One whole NATLayer without layer scale (reference backbone/dinat.py:90-97): LN1 -> qkv -> neighbourhood attention ->
    proj (+x) -> LN2 -> fc1 + GELU -> fc2 (+x), the same kernel sequence as ops.SwinBlockFn with uenc_na2d in the middle.
    x is the fp32 residual stream (B, H, W, C), H, W >= ks * dilation.  dp: as in SwinBlockFn (DropPath scales, training mode)."""

    @staticmethod
    def forward(ctx, x, nH, ks, dilation, scale, dp, g1, b1, wqkv, bqkv, rpb, wproj, bproj, g2, b2, w1, bb1, w2, bb2):
        B, H, W, C = x.shape
        s1, s2 = dp if dp is not None else (None, None)
        M = B * H * W
        x2 = x.reshape(M, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        xn, _, st1 = K.layernorm_fwd(x2, g1.detach(), b1.detach(), out_dtype=BF16)
        qkv = K.gemm_nt(xn, CACHE.mat(wqkv), bias=None if bqkv is None else bqkv.detach())
        rp = rpb.detach().float().contiguous()
        attn, lse = K.na2d_fwd(qkv.view(B, H, W, 3 * C), rp, nH, ks, dilation, scale)
        x1 = _branch_gemm(attn.view(M, C), CACHE.mat(wproj), bproj.detach(), x2, C, s1)
        xn2, _, st2 = K.layernorm_fwd(x1, g2.detach(), b2.detach(), out_dtype=BF16)
        pre = torch.empty((M, w1.shape[0]), dtype=BF16, device=x.device)
        h = K.gemm_nt(xn2, CACHE.mat(w1), bias=bb1.detach(), epilogue=K.EPI_GELU, aux_out=pre)
        x2o = _branch_gemm(h, CACHE.mat(w2), bb2.detach(), x1, C, s2)
        ctx.dp = (s1, s2)
        ctx.save_for_backward(x2, st1, xn, qkv, rp, lse, attn, x1, st2, xn2, pre, h,
                              g1, b1, wqkv, bqkv, rpb, wproj, bproj, g2, b2, w1, bb1, w2, bb2)
        ctx.geom = (B, H, W, C, nH, ks, dilation, scale)
        return x2o.view(B, H, W, C)

    @staticmethod
    def backward(ctx, dxo):
        (x2, st1, xn, qkv, rp, lse, attn, x1, st2, xn2, pre, h,
         g1, b1, wqkv, bqkv, rpb, wproj, bproj, g2, b2, w1, bb1, w2, bb2) = ctx.saved_tensors
        B, H, W, C, nH, ks, dilation, scale = ctx.geom
        M = B * H * W
        d2 = dxo.reshape(M, C)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        if d2.dtype != F32:
            d2 = d2.float()
        train = wqkv.requires_grad
        s1, s2 = ctx.dp
        d2h = _twin(d2)
        if d2h is None:
            d2h = K.cast_bf16(d2)
        if K.EXACT:
            d2h, s2 = _scaled_rows(d2h, s2), None
        dh = _branch_dgrad(d2h, CACHE.mat_t(w2), s2, epilogue=K.EPI_MUL_DGELU, aux=pre)
        if train:
            _branch_wgrad(d2h, h, grad_buf(w2), grad_buf(bb2), (w2, bb2), s2)
        dxn2 = K.gemm_nt(dh, CACHE.mat_t(w1))
        if train:
            _tn(dh, xn2, grad_buf(w1), grad_buf(bb1), (w1, bb1))
        tw = []
        dx1 = _ln_bwd(dxn2, x1, st2, g2.detach(), dres=d2,
                              dgamma=grad_buf(g2) if train else None, dbeta=grad_buf(b2) if train else None, twin=tw)
        dx1h = tw[0]
        if K.EXACT:
            dx1h, s1 = _scaled_rows(dx1h, s1), None
        dattn = _branch_dgrad(dx1h, CACHE.mat_t(wproj), s1)
        if train:
            _branch_wgrad(dx1h, attn.view(M, C), grad_buf(wproj), grad_buf(bproj), (wproj, bproj), s1)
        dqkv = K.na2d_bwd(qkv.view(B, H, W, 3 * C), rp, attn, dattn.view(B, H, W, C), lse, nH, ks, dilation, scale,
                          grad_buf(rpb) if (train and rpb.requires_grad) else None)
        dqkv2 = dqkv.view(M, 3 * C)
        dxn = K.gemm_nt(dqkv2, CACHE.mat_t(wqkv))
        if train:
            _tn(dqkv2, xn, grad_buf(wqkv), None if bqkv is None else grad_buf(bqkv), (wqkv, bqkv))
        tw = []
        dx = _ln_bwd(dxn, x2, st1, g1.detach(), dres=dx1,
                             dgamma=grad_buf(g1) if train else None, dbeta=grad_buf(b1) if train else None, twin=tw)
        _register_twin(dx, tw[0])
        if train:
            _tn_notify(g1, b1, rpb, g2, b2)
        return (dx.view(B, H, W, C),) + (None,) * 18


def nat_layer(x, nH, ks, dilation, scale, params: Sequence[torch.Tensor], dp=None):
    return NATLayerFn.apply(x, nH, ks, dilation, scale, dp, *params)


class ConvS2Fn(torch.autograd.Function):
    """3x3 stride-2 padding-1 convolution on a channels-last map (ConvTokenizer / ConvDownsampler, reference backbone/dinat.py:17-45)
    as patch gather + MFMA GEMM: x (B, H, W, Cin) -> (B, Ho, Wo, Cout) fp32, Ho = ceil(H / 2).  The patch matrix is gathered with
    strided slices (data movement only); forward, input gradient and weight gradient are the library's GEMMs."""

    @staticmethod
    def _patches(x16, Ho, Wo, Kp):
        B, H, W, C = x16.shape
        xp = F.pad(x16, (0, 0, 1, 2 * Wo - W, 1, 2 * Ho - H))            # pad 1 left / top; right / bottom up to the last tap
        taps = [xp[:, dy:dy + 2 * Ho:2, dx:dx + 2 * Wo:2, :] for dy in range(3) for dx in range(3)]
        if Kp > 9 * C:
            taps.append(x16.new_zeros((B, Ho, Wo, Kp - 9 * C)))
        return torch.cat(taps, dim=-1).reshape(B * Ho * Wo, Kp)

    @staticmethod
    def _wmat(weight, Kp, transposed):
        def make():
            w = weight.detach().permute(0, 2, 3, 1).reshape(weight.shape[0], -1)
            w = F.pad(w, (0, Kp - w.shape[1], 0, -weight.shape[0] % 8)).contiguous()
            return K.cast_transpose_bf16(w) if transposed else K.cast_bf16(w)
        return CACHE._get(weight, "s2t" if transposed else "s2", make)

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, H, W, C = x.shape
        Co = weight.shape[0]
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        Kp = -(-9 * C // 8) * 8
        x16 = x if x.dtype == BF16 else x.to(BF16)
        # patch matrix by the HIP gather when the channel count allows 16-byte pieces, else by strided slices (the 3-channel image)
        col = K.im2col3x3_s2(x16.contiguous()) if C % 8 == 0 else ConvS2Fn._patches(x16, Ho, Wo, Kp)
        Np = -(-Co // 8) * 8
        out = K.gemm_nt(col, ConvS2Fn._wmat(weight, Kp, False), bias=_bias_pad(bias, Np), out_dtype=F32)
        ctx.save_for_backward(col, weight, bias)
        ctx.shape = (B, H, W, C, Ho, Wo, Kp, Np)
        ctx.in_dtype = x.dtype
        return out[:, :Co].reshape(B, Ho, Wo, Co) if Np != Co else out.view(B, Ho, Wo, Co)

    @staticmethod
    def backward(ctx, dy):
        col, weight, bias = ctx.saved_tensors
        B, H, W, C, Ho, Wo, Kp, Np = ctx.shape
        Co = weight.shape[0]
        dy2 = dy.reshape(B * Ho * Wo, Co)
        dy2 = K.cast_bf16(dy2.float().contiguous()) if dy2.dtype != BF16 else dy2.contiguous()
        if Np != Co:
            dy2 = F.pad(dy2, (0, Np - Co))
        dx = None
        if ctx.needs_input_grad[0]:
            if C % 8 == 0:
                dcol = K.gemm_nt(dy2, ConvS2Fn._wmat(weight, Kp, True))                          # (M, 9C) bf16
                dx = K.col2im3x3_s2(dcol, B, H, W, C)
                if ctx.in_dtype != F32:
                    dx = dx.to(ctx.in_dtype)
            else:
                dcol = K.gemm_nt(dy2, ConvS2Fn._wmat(weight, Kp, True), out_dtype=F32).view(B, Ho, Wo, Kp)
                dxp = dcol.new_zeros((B, 2 * Ho + 2, 2 * Wo + 2, C))
                for t in range(9):                                        # adjoint of the strided gather
                    dyy, dxx = divmod(t, 3)
                    dxp[:, dyy:dyy + 2 * Ho:2, dxx:dxx + 2 * Wo:2, :] += dcol[..., t * C:(t + 1) * C]
                dx = dxp[:, 1:1 + H, 1:1 + W, :].to(ctx.in_dtype)
        if weight.requires_grad:
            dw = torch.zeros((Np, Kp), dtype=F32, device=dy.device)
            db = torch.zeros((Np,), dtype=F32, device=dy.device) if bias is not None else None
            K.gemm_tn(dy2, col, dw, db)
            grad_buf(weight).add_(dw[:Co, :9 * C].view(Co, 3, 3, C).permute(0, 3, 1, 2))
            if bias is not None:
                grad_buf(bias).add_(db[:Co])
            _tn_notify(weight, bias)
        return dx, None, None


def conv3x3_s2(x_nhwc, weight, bias=None):
    """x (B, H, W, Cin) channels-last -> (B, ceil(H/2), ceil(W/2), Cout) fp32."""
    return ConvS2Fn.apply(x_nhwc, weight, bias)


class PatchMergeLnFn(torch.autograd.Function):
    """PatchMerging's pad + 2x2 strided gather + concat + LayerNorm(4C) (reference backbone/swin.py:311-334) as one kernel each way:
    x (B, H, W, C) fp32 -> (B, ceil(H/2) * ceil(W/2), 4C) bf16, the operand of the 4C -> 2C reduction GEMM."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xc = x if x.is_contiguous() else x.contiguous()
        y, stats = K.patch_merge_ln_fwd(xc, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(xc, stats, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma, beta = ctx.saved_tensors
        train = gamma.requires_grad
        dx = _pm_ln_bwd(dy if dy.is_contiguous() else dy.contiguous(), x, stats, gamma.detach(),
                                  grad_buf(gamma) if train else None, grad_buf(beta) if train else None)
        if train:
            _tn_notify(gamma, beta)          # (held until the parked dgamma / dbeta partials are summed)
        return dx, None, None, None


def patch_merge_ln(x_bhwc, gamma, beta, eps: float = 1e-5):
    return PatchMergeLnFn.apply(x_bhwc, gamma, beta, eps)
