#!/usr/bin/env python3
"""What a plain elementwise pass reaches on this GPU at BatchNorm-sized tensors (yardstick for bn_* and Adam)."""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (8, 24, 47, 94, 189, 400, 1600):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device="cuda"); y = torch.empty_like(x); z = torch.randn(n, device="cuda")
    us = t(lambda: torch.add(x, 1.0, out=y))
    us3 = t(lambda: torch.add(x, z, out=y))
    usr = t(lambda: x.sum())
    print("%5d MB: y=x+1 %.1f us %.2f TB/s | y=x+z %.1f us %.2f TB/s | sum(x) %.1f us %.2f TB/s" % (mb, us, 2 * n * 4 / us / 1e6, us3, 3 * n * 4 / us3 / 1e6, usr, n * 4 / usr / 1e6))
ps = [torch.randn(s, device="cuda", requires_grad=True) for s in [2359296] * 8 + [589824] * 8 + [147456] * 8 + [36864] * 8 + [512] * 60]
for p in ps: p.grad = torch.randn_like(p)
opt = torch.optim.Adam(ps, 1e-4, fused=True)
us = t(lambda: opt.step())
nb = sum(p.numel() for p in ps) * 4
print("fused Adam over %.1f M params: %.1f us, %.2f TB/s (7 streams)" % (nb / 4e6, us, 7 * nb / us / 1e6))
