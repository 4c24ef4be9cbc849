import os, sys, torch
sys.path.insert(0, os.getcwd())
import gsdd_amd
from gsdd_amd import ops
H = 16
for (B, L, sc) in [(2, 64, 1.5), (1, 1024, 1.5), (1, 1024, 4.0), (2, 4096, 1.5)]:
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, H, L, 4, generator=g) * sc
    k = torch.randn(B, H, L, 4, generator=g) * sc
    v = torch.randn(B, H, L, 4, generator=g)
    att = torch.softmax((q.double() @ k.double().transpose(-1, -2)) * 0.5, dim=-1)
    want = (att @ v.double()).permute(0, 2, 1, 3).reshape(B * L, H * 4)
    hm = lambda z: z.permute(1, 0, 2, 3).reshape(H, B * L, 4).contiguous().cuda()
    out = torch.empty((B * L, H * 4), device="cuda")
    ops.d3pm_attention(hm(q), hm(k), hm(v), B, L, H, out, ws=ops.d3pm_attention_workspace(B, L, H, 'cuda'))
    err = (out.cpu().double() - want).abs()
    print(f"B={B} L={L} scale={sc}: max abs err {err.max().item():.3e}  mean {err.mean().item():.3e}  (|out| max {want.abs().max().item():.2f})")
