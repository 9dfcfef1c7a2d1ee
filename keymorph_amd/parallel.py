"""Data-parallel training over image pairs: one process per GPU, RCCL over xGMI.

The reference's only multi-GPU mechanism is nn.DataParallel around the backbone
(scripts/run.py:390).  Pairwise registration is embarrassingly parallel over pairs,
so here every rank runs the WHOLE pair pipeline on its own pairs; the only exchange is one
all-reduce (sum) of a single flat fp32 gradient bucket per step -- 16 MB for
TruncatedUNet3D(512 kp), far below one xGMI link's per-step capacity -- followed by one fused
Adam launch on the same flat buffer (identical update on every rank).
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .ops import _p, _stream


def init_distributed():
    """(rank, local_rank, world).  backend 'nccl' (= RCCL) on GPUs, gloo on CPU.
    Test hooks for a 1-GPU box: KEYMORPH_DIST_BACKEND=gloo and KEYMORPH_SHARE_GPU=1 let several ranks share device 0
    (RCCL refuses two ranks on one GPU); never set in production."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("KEYMORPH_SHARE_GPU") == "1":
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("KEYMORPH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():      # every backend: the HIP library launches on the current device
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_indices(num_items: int, rank: int, world: int) -> List[int]:
    """Contiguous block partition of pair / subject indices (remainder to the low ranks)."""
    base, rem = divmod(num_items, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


class FlatParams:
    """Re-homes parameters (and their .grad) into two flat fp32 buffers: one all-reduce, one Adam.

    Gradients are GATHERED, not accumulated: `zero_grad()` drops every `.grad` (no 64 MB memset), autograd then simply
    assigns each parameter's gradient tensor (no read-modify-write `add` launch per parameter: 36 of them per step for
    TruncatedUNet3D), and the first access to `.grad` afterwards copies them into the flat buffer with one multi-tensor
    copy and re-points every `p.grad` at its slice (`gather()`; `allreduce_grads` and `FusedAdam.step` call it).  Gradient
    accumulation over several backward passes works as long as `gather()` / `.grad` is read between them."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self._grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self._views = []
        o = 0
        for p in self.params:
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + k].view_as(p.data)
            v = self._grad[o:o + k].view_as(p.data)
            p.grad = v
            self._views.append(v)
            o += k
        self.numel = n

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def gather(self) -> torch.Tensor:
        """Bring the flat gradient buffer up to date with every parameter's `.grad` (one multi-tensor copy, one multi-tensor
        zero for parameters that received no gradient) and re-point each `p.grad` at its slice.  Returns the flat buffer.
        CONTRACT: call this (or read `.grad`, which calls it) after EVERY backward pass; a reference to `flat.grad` or
        `p.grad` taken before a backward pass is stale afterwards, and `_grad` must not be read directly."""
        src, dst, zero = [], [], []
        for p, v in zip(self.params, self._views):
            g = p.grad
            if g is None:
                zero.append(v)
            elif g.data_ptr() != v.data_ptr():
                src.append(g.detach().reshape(v.shape).to(v.dtype))
                dst.append(v)
            p.grad = v
        with torch.no_grad():
            if zero:
                torch._foreach_zero_(zero)
            if dst:
                torch._foreach_copy_(dst, src)
        return self._grad

    @property
    def grad(self) -> torch.Tensor:
        """`gather()`: the flat gradient buffer, up to date with every parameter's `.grad` (has that side effect)"""
        return self.gather()

    def broadcast(self, src: int = 0):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.flat, src)

    def allreduce_grads(self) -> float:
        """sum over ranks (RCCL); returns the scale (1/world) the optimizer applies."""
        g = self.gather()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0


class FusedAdam:
    """torch.optim.Adam (defaults) as one HIP launch over FlatParams."""

    def __init__(self, flat: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.t = 0

    def state_dict(self):
        """torch.optim.Adam's layout (state[i] = {step, exp_avg, exp_avg_sq}, one param group), so checkpoints are
        interchangeable with the reference's (scripts/run.py:588-602, script_utils.py:59-81)."""
        state, o = {}, 0
        for i, p in enumerate(self.flat.params):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": self.m[o:o + k].view_as(p).clone(),
                        "exp_avg_sq": self.v[o:o + k].view_as(p).clone()}
            o += k
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.flat.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        assert len(groups) == 1 and len(groups[0]["params"]) == len(self.flat.params), "parameter list mismatch"
        g = groups[0]
        assert not g.get("amsgrad", False) and not g.get("weight_decay", 0), "plain Adam only"
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        o, steps = 0, set()
        for i, p in zip(g["params"], self.flat.params):
            k = p.numel()
            st = sd["state"].get(i)
            if st is None:                                   # parameter never stepped
                self.m[o:o + k].zero_(); self.v[o:o + k].zero_()
            else:
                self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            o += k
        assert len(steps) <= 1, "per-parameter step counts differ"
        self.t = steps.pop() if steps else 0

    def step(self, grad_scale: float = 1.0):
        self.t += 1
        lib = _lib.load()
        check(lib.kmh_adam_step(_p(self.flat.flat), _p(self.flat.gather()), _p(self.m), _p(self.v), self.flat.numel,
                                self.lr, self.betas[0], self.betas[1], self.eps, self.t, grad_scale, _stream()),
              "kmh_adam_step")


def gather_group_points(local_points: torch.Tensor, num_subjects: int) -> torch.Tensor:
    """Groupwise registration sharded by `shard_indices`: rank r holds the keypoints (n_r, K, 3) of ITS contiguous block
    of subjects; returns the full (num_subjects, K, 3) set in subject order on every rank.  Blocks differ by at most
    one subject, so the shorter ones are padded to the longest for ONE all-gather (RCCL) and trimmed afterwards."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        assert local_points.shape[0] == num_subjects
        return local_points
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [len(shard_indices(num_subjects, r, world)) for r in range(world)]
    assert local_points.shape[0] == counts[rank], "this rank's block does not match shard_indices()"
    m = max(counts)
    pad = local_points.new_zeros((m,) + tuple(local_points.shape[1:]))
    pad[: counts[rank]] = local_points
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad.contiguous())
    return torch.cat([o[:c] for o, c in zip(outs, counts)], dim=0)


def allgather_points(points: torch.Tensor) -> torch.Tensor:
    """Groupwise registration (config 5): every rank extracts its subjects' keypoints, one all-gather
    of (n_local, K, 3) makes the (N, K, 3) set available everywhere (6 KB per subject)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return points
    outs = [torch.empty_like(points) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, points.contiguous())
    return torch.cat(outs, dim=0)
