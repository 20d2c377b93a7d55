"""K7 lab: sweep the split-B factor at the cfg2 arena shape (needs a lib built with -DDPK_LAB_ENV)."""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
os.environ["DPK_LIB_PATH"] = os.path.join(here, "libdpk_lab_env.so")
import torch
from dp_timeautodiff_amd import _native as nat
sys.path.insert(0, os.path.join(here, ".."))
from kbench import timeit
for (B, P) in ((1024, 824448), (64, 649408), (4096, 2830272)):
    base = torch.randn(B, P, device="cuda"); clip = torch.rand(B, device="cuda")
    ws = torch.empty(64 * P, device="cuda")
    summed = torch.zeros(P, device="cuda"); grad = torch.zeros(P, device="cuda")
    v = nat.FlatView(base, 0, P)
    for ns in (1, 2, 3, 4, 5, 6, 8, 12, 16, 32):
        if ns > B // 8: continue
        os.environ["DPK_K7_NSPLIT"] = str(ns)
        ms = timeit(lambda: nat.clip_reduce_noise(v, clip, None, None, 1, 0, 1.0, 1.0 / B, summed, grad, B, ws))
        print(f"B={B} P={P} nsplit={ns:2d}: {ms*1e3:8.1f} us  {B*P*4/ms/1e6:7.1f} GB/s", flush=True)
    del base, ws
