"""Multi-GPU layout: one process per GPU (torch.distributed.run), RCCL over xGMI (`backend="nccl"` is RCCL on ROCm).

Sampling: the batch of independent clips is split across ranks, no data-path collective ("replicas").
Noise is keyed by the GLOBAL sample row (Philox counter = global_row * row_quads + quad), so the tokens a clip gets do not depend
on the GPU count.  Training: data parallel; parameters and buffers are broadcast from rank 0 once (what DDP does when it wraps a
module), gradients are averaged by all-reduce in few large buckets issued while the backward is still running (xGMI is
point-to-point: a ring all-reduce is bound by one link, so large messages, few launches)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the job launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).  Binds this
    process to cuda:LOCAL_RANK BEFORE the process group exists (with nccl every rank would otherwise land on cuda:0).  A backend
    that cannot initialise raises with the rendezvous it tried, rather than hanging later in the first collective."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("GSDD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        if torch.cuda.is_available():
            n = torch.cuda.device_count()
            if backend == "nccl" and local >= n:
                raise RuntimeError(f"LOCAL_RANK {local} but only {n} GPU(s) visible: one process per GPU is the layout")
            torch.cuda.set_device(local % max(n, 1))
        try:
            dist.init_process_group(backend)
        except Exception as e:
            raise RuntimeError(f"torch.distributed backend '{backend}' failed to initialise (WORLD_SIZE={world}, RANK="
                               f"{os.environ.get('RANK')}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}, MASTER_PORT="
                               f"{os.environ.get('MASTER_PORT')}): {e}") from e
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_batch(global_batch, world, rank):
    """Contiguous split of `global_batch` clips: -> (first global sample index, local count)."""
    base, rem = divmod(global_batch, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_tokens(tok):
    """Optional final gather of (B_local, L) int64 tokens to every rank (equal shard sizes assumed padded by caller)."""
    if world_size() == 1:
        return tok
    parts = [torch.empty_like(tok) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, tok.contiguous())
    return torch.cat(parts, 0)


@torch.no_grad()
def broadcast_module(module, src=0):
    """Every parameter and buffer takes rank `src`'s value (DDP's start-up broadcast): ranks that were not seeded identically
    would otherwise average gradients taken at different weights and drift apart silently.  Also clears packed-weight caches."""
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_floating_point() or t.dtype in (torch.int64, torch.int32, torch.bool, torch.uint8):
            dist.broadcast(t.data, src)
    for m in module.modules():
        if hasattr(m, "_packed"):
            m._packed = None


@torch.no_grad()
def broadcast_buffers(module, src=0):
    """Rank `src`'s buffers to every rank -- what DistributedDataParallel(broadcast_buffers=True, the default the reference trains
    under) does at the start of every forward: the D3PM importance-sampling statistics `Lt_history` / `Lt_count`
    (diffusion_transformer.py:425-433) are rank 0's on every rank, so all ranks draw timesteps from the same distribution."""
    if world_size() == 1:
        return
    for t in module.buffers():
        if t.is_floating_point() or t.dtype in (torch.int64, torch.int32):
            dist.broadcast(t.data, src)


def set_rank_noise_rows(diffusion_model, local_batch):
    """Data parallel: this rank's clips are rows rank * B .. of the global batch, so the ranks draw different q_sample / Gumbel noise
    rows and a world-size-N step equals the single-process step on the concatenated batch."""
    if world_size() > 1:
        diffusion_model.row_offset = dist.get_rank() * int(local_batch)


def assert_same_parameters(module, what="parameters"):
    """Cheap drift check: the fp64 sum of all parameters must agree across ranks."""
    if world_size() == 1:
        return
    s = torch.zeros(1, dtype=torch.float64, device=next(module.parameters()).device)
    for p in module.parameters():
        s += p.detach().double().sum()
    lo, hi = s.clone(), s.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if lo.item() != hi.item():
        raise RuntimeError(f"{what} differ across ranks (checksum {lo.item()} .. {hi.item()}): broadcast them first")


class GradReducer:
    """Mean of gradients over the data-parallel group, in buckets that are issued asynchronously while later backward kernels are
    still being enqueued.  `add(t)` queues an in-place all-reduce of the flat fp32 tensor `t` (a slice of the trainer's gradient
    arena, or a concatenation made by the caller); `finish()` waits for all of them and divides by the world size.  Records the
    time from the first issue to the last completion (`last_ms`, HIP events on the current stream) so that the share of a step
    spent in the exchange is a measured number."""

    def __init__(self):
        self.handles, self.tensors = [], []
        self.last_ms, self.last_exposed_ms, self.last_bytes, self.last_buckets = 0.0, 0.0, 0, 0
        self._ev = None

    def active(self):
        # GSDD_REDUCER_FORCE: also with a one-rank group (a rehearsal of the exchange on hardware where only one GPU is to be had:
        # RCCL initialises, the buckets go out on its stream during the backward, the compute stream waits for them)
        return world_size() > 1 or (os.environ.get("GSDD_REDUCER_FORCE") is not None and dist.is_available() and dist.is_initialized())

    def add(self, t):
        if not self.active() or t.numel() == 0:
            return
        if not self.handles and t.is_cuda:
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._ev[0].record()
        self.tensors.append(t)
        self.handles.append(dist.all_reduce(t, async_op=True))

    def finish(self):
        if not self.handles:
            return
        mid = None
        if self._ev is not None:
            mid = torch.cuda.Event(enable_timing=True)     # everything the backward enqueued is before this point
            mid.record()
        for h in self.handles:
            h.wait()                                       # the current stream waits for the collective; the host does not block
        w = float(dist.get_world_size())
        for t in self.tensors:
            t.div_(w)
        self.last_bytes = sum(t.numel() * t.element_size() for t in self.tensors)
        self.last_buckets = len(self.tensors)
        if self._ev is not None:
            self._ev[1].record()
            self._pending = (self._ev[0], mid, self._ev[1])
        self.handles, self.tensors, self._ev = [], [], None

    def stats(self):
        """{'allreduce_span_ms', 'allreduce_exposed_ms', 'allreduce_mib', 'allreduce_buckets'} of the last exchange ({} before the
        first one).  span = first bucket issued -> averaged gradients ready (overlaps the rest of the backward); exposed = the part
        after the backward's last kernel, i.e. what the exchange adds to the step."""
        if getattr(self, "_pending", None) is None:
            return {}
        span, exposed = self.elapsed_ms()
        return {"allreduce_span_ms": span, "allreduce_exposed_ms": exposed, "allreduce_mib": self.last_bytes / 2 ** 20,
                "allreduce_buckets": self.last_buckets}

    def elapsed_ms(self):
        """(span, exposed) of the last finished exchange in ms: first issue -> averaged gradients ready, and the part of it after the
        backward's last kernel (what the step actually waits for).  Synchronises on the closing event."""
        ev = getattr(self, "_pending", None)
        if ev is None:
            return 0.0, 0.0
        ev[2].synchronize()
        self.last_ms = ev[0].elapsed_time(ev[2])
        self.last_exposed_ms = ev[1].elapsed_time(ev[2])
        return self.last_ms, self.last_exposed_ms
