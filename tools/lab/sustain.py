"""Sustained single-kernel loop (for sampling clocks/power with rocm-smi beside it)."""
import sys, time; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch
from dp_timeautodiff_amd import _native as nat
which = sys.argv[1]; secs = float(sys.argv[2])
if which == "k1":
    B, T, in_f, out_f = 2048, int(sys.argv[3]), 512, 1024
    A = torch.randn(B, T, in_f, device="cuda"); G = torch.randn(B, T, out_f, device="cuda")
    n = in_f * out_f
    base = torch.zeros(B, n + 64, device="cuda"); part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
    vw = nat.FlatView(base, 0, n)
    fn = lambda: nat.linear_grad_sample(A, G, vw, None, part, 0, float(B), ld_w=in_f)
    unit = B * n * 4
elif which == "k7":
    B, P = 2048, 2830272
    base = torch.randn(B, P, device="cuda"); clip = torch.rand(B, device="cuda")
    ws = torch.empty(max(nat.clip_reduce_workspace(B, P) // 4, 4), device="cuda")
    summed = torch.zeros(P, device="cuda"); grad = torch.zeros(P, device="cuda"); v = nat.FlatView(base, 0, P)
    fn = lambda: nat.clip_reduce_noise(v, clip, None, None, 1, 0, 1.0, 1.0 / B, summed, grad, B, ws)
    unit = B * P * 4
elif which == "fill":
    base = torch.empty(2048 * 524288, device="cuda")
    fn = lambda: base.fill_(1.0)
    unit = base.numel() * 4
t_end = time.time() + secs
while time.time() < t_end:
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{which} {ms*1e3:8.1f} us  {unit/ms/1e6:7.1f} GB/s", flush=True)
