"""world_size-2 gloo test (CPU) of the N>1 path: batch sharding + global-row noise keying give every clip the
same tokens as the single-process run.  The per-rank compute here is the CPU oracle (the HIP path needs a GPU);
what is under test is the host-side sharding / keying / gather logic that bench.py --gpus N uses."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

STEPS = 4


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import gsdd_amd
    from gsdd_amd.parallel import gather_tokens, init_distributed, shard_batch
    from oracle import d3pm as od
    from tests.conftest import load_golden
    r, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    sd, a, cfg = load_golden("d3pm_L64")
    start, count = shard_batch(cfg["B"], world, rank)
    cond = torch.from_numpy(a["step_cond"])[start:start + count]
    with torch.no_grad():
        tok = od.sample(count, cfg["L"], cond, torch.zeros_like(cond), sd, cfg["guidance"], cfg["noise_seed"],
                        row0=start * cfg["L"], steps=STEPS)
    allt = gather_tokens(tok)
    if rank == 0:
        out_q.put(allt.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reproduce_single_process_tokens(golden):
    _, a, cfg = golden("d3pm_L64")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, a["loop_trace"][STEPS - 1])


def test_shard_batch_partitions():
    from gsdd_amd.parallel import shard_batch
    for B in (1, 7, 16, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_batch(B, w, r) for r in range(w)]
            assert sum(c for _, c in spans) == B
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
