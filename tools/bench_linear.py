#!/usr/bin/env python3
"""Row GEMMs of the D3PM training step (M = 16 * 4096 rows) through gsdd_gemm: time and bytes/s per shape and prologue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsdd_amd
from gsdd_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M, L, D = 16 * 4096, 4096, 64
dev = "cuda"
x = torch.randn((M, D), device=dev)
stats = torch.empty((M, 2), device=dev)
g = torch.randn((100, 2 * D), device=dev)
t2 = torch.full((16,), 50, dtype=torch.int64, device=dev)
ops.row_stats(x, stats)
for cin, cout in [(64, 64), (64, 128), (64, 192), (64, 256), (256, 64), (192, 64), (128, 64)]:
    for ln in ([False, True] if cin == 64 else [False]):
        for mode in ([0, 2] if cout == 192 else [0]):
            w = torch.randn((cout, cin), device=dev) * 0.05
            b = torch.randn((cout,), device=dev)
            a = torch.randn((M, cin), device=dev)
            o = torch.empty((M, cout), device=dev)
            lnarg = (stats, g.view(-1), g.view(-1)[D:], t2, 2 * D) if ln else None
            ms = timeit(lambda: ops.linear(a, w, o, bias=b, ln=lnarg, rows_per_batch=L, out_mode=mode))
            by = 4.0 * M * (cin + cout)
            print(f"linear {cin:3d}->{cout:3d} ln={int(ln)} mode={mode}: {ms * 1e3:6.1f} us  {by / ms / 1e6:5.0f} GB/s  {2.0 * M * cin * cout / ms / 1e9:5.1f} TFLOP/s")
