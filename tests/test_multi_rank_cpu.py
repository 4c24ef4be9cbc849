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


def _dp_host_worker(rank, world, port, out_q):
    """Host side of the data-parallel training path on CPU tensors (gloo): start-up broadcast, bucketed gradient averaging, the
    drift check, the round-robin deal of batches and the all-reduced epoch losses."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import gsdd_amd  # noqa: F401
    import src  # noqa: F401
    from gsdd_amd.parallel import GradReducer, assert_same_parameters, broadcast_module, init_distributed
    from src.models.metrics.loss import ComputeLosses
    from src.tasks.runner import Trainer
    r, w = init_distributed("gloo")
    torch.manual_seed(100 + rank)                                  # ranks seeded differently on purpose
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    try:
        assert_same_parameters(net)
        drift_seen = False
    except RuntimeError:
        drift_seen = True
    broadcast_module(net)
    assert_same_parameters(net)
    red = GradReducer()
    g1, g2 = torch.full((5,), float(rank + 1)), torch.full((3,), 10.0 * (rank + 1))
    red.add(g1)
    red.add(g2)                                                    # two buckets in flight
    red.finish()
    tr = Trainer(max_epochs=1)
    dealt = [i for i, _ in tr._batches(range(7))], [b for _, b in tr._batches(range(7))]
    cl = ComputeLosses(loss_dict={"l_dummy": 1.0})
    cl.update({"losses": torch.tensor(float(rank + 1))})
    out_q.put((rank, drift_seen, net[0].weight.detach().numpy().copy(), g1.numpy(), g2.numpy(), red.last_buckets, dealt,
               float(cl.compute()["total"])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_host_logic():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_host_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, d0, w0, a0, b0, n0, deal0, tot0), (_, d1, w1, a1, b1, n1, deal1, tot1) = res
    assert d0 and d1                                               # both ranks notice the drift before the broadcast
    assert np.array_equal(w0, w1)                                  # rank 0's weights everywhere
    assert np.allclose(a0, 1.5) and np.allclose(a1, 1.5) and np.allclose(b0, 15.0) and np.allclose(b1, 15.0) and n0 == n1 == 2
    assert deal0 == ([0, 1, 2], [0, 2, 4]) and deal1 == ([0, 1, 2], [1, 3, 5])      # whole rounds only: batch 6 is dropped
    assert tot0 == tot1 == 1.5                                     # epoch losses are all-reduced means
