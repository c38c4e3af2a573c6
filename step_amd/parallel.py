"""Data-parallel training helpers: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

The ST-GCN stack shards over the batch (sequences are independent); the only
exchange step of a training iteration is ONE all-reduce of the flat gradient bucket
(716 292 fp32 = 2.87 MB for the classifiers) between backward and optimizer.step().
Inference and synthesis use no collective at all.  BatchNorm statistics stay local to a
rank by default (standard DDP semantics); the reference itself is single-device, and
``sync_batchnorm(True)`` reproduces ITS statistics (one batch over all ranks) at the price of one small fp64
all-reduce per BatchNorm in forward and one in backward.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); no-op when alone."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (1-GPU box): STEP_DIST_BACKEND=gloo lets several ranks share one card,
    # STEP_LOCAL_DEVICE pins every rank to that card
    backend = backend or os.environ.get("STEP_DIST_BACKEND")
    if "STEP_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["STEP_LOCAL_DEVICE"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_batch(n_total, rank, world):
    """Contiguous [start, stop) slice of a global batch for this rank (remainder to low ranks)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _host_staged(group=None):
    """gloo moves host memory: device tensors are staged through the CPU for it (rehearsal / CPU tests);
    RCCL ('nccl') reduces device buffers in place over xGMI."""
    return dist.get_backend(group) == "gloo"


def _all_reduce_sum(flat, group=None):
    if flat.is_cuda and _host_staged(group):
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


def _broadcast(flat, src, group=None):
    if flat.is_cuda and _host_staged(group):
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src=src, group=group)


class GradBucket:
    """All parameters' gradients as ONE persistent flat fp32 buffer; every ``p.grad`` is a view into it, so the
    step's single all-reduce runs on the buffer in place - no ``cat`` before it, no copy back after it.

    Use ``bucket.zero_grad()`` instead of ``optimizer.zero_grad()`` (one memset; ``set_to_none=True`` would drop
    the views - they are re-attached on the next call if that happened): autograd then accumulates each
    gradient into its slice in place."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self._views = []

    def _attach(self):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
            self._views = []
            off = 0
            for p in self.params:
                if p.dtype != torch.float32:
                    raise TypeError("GradBucket holds fp32 parameters only")
                self._views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        for p, v in zip(self.params, self._views):
            if p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                else:
                    v.zero_()               # dropped by zero_grad(set_to_none=True): never a stale averaged gradient
                p.grad = v

    def zero_grad(self):
        self._attach()
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return
        self._attach()                       # gradients made outside the views (first step) are copied in once
        _all_reduce_sum(self.flat, group)
        self.flat.div_(dist.get_world_size(group))


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers: one broadcast per dtype."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for dtype in sorted(by_dtype, key=str):
        ts = by_dtype[dtype]
        flat = torch.cat([t.reshape(-1) for t in ts])
        _broadcast(flat, src, group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def sync_batchnorm(enabled=True, group=None):
    """Train-mode BatchNorm of every st_gcn / st_gctn block and of `data_bn` over the batch of ALL ranks: with it
    W ranks x N/W sequences compute what the reference's single device computes on N (same normalisation, running
    statistics and gradients).  Per BatchNorm the ranks all-reduce 2*G + 1 doubles (SUM x, SUM x^2, element count)
    in forward and (SUM g, SUM g*x, count) in backward; dgamma / dbeta stay local and are averaged with the other
    gradients by GradBucket.all_reduce_mean.  BatchNorms whose inputs exist at the same time share one all-reduce
    (tcn[0] + residual[1] after the graph conv; tcn[3] + residual[1] in the backward tail); the hybrid head's
    BatchNorm1d on feature rows (data_bn2) is covered the same way (step_bn_rows_stats / _apply)."""
    from . import ops
    ops.set_sync_bn(enabled, group)
