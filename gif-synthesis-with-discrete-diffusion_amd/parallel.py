"""Multi-GPU layout of the sampling path: one process per GPU, the batch of independent clips is split across
ranks, no data-path collective ("replicas").  Noise is keyed by the GLOBAL sample row
(Philox counter = global_row * row_quads + quad), so the tokens a clip gets do not depend on the GPU count."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the job launched by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* from the env)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def shard_batch(global_batch, world, rank):
    """Contiguous split of `global_batch` clips: -> (first global sample index, local count)."""
    base, rem = divmod(global_batch, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_tokens(tok):
    """Optional final gather of (B_local, L) int64 tokens to every rank (equal shard sizes assumed padded by caller)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tok
    parts = [torch.empty_like(tok) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, tok.contiguous())
    return torch.cat(parts, 0)
