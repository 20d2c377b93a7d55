"""Replay K1..K7 of one config a few times (for rocprofv3 --pmc passes / kernel traces).
usage: python3 tools/hotpath_replay.py cfg2 [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dp_timeautodiff_amd import _native, workloads
from dp_timeautodiff_amd.grad_sample import GradSampleModule
from dp_timeautodiff_amd.optimizer import DPOptimizer

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
_native.load()
wl = workloads.build(cfg, dev, batch=batch)
gsm = GradSampleModule(wl.model)
opt = DPOptimizer(torch.optim.Adam(wl.model.parameters()), noise_multiplier=1.0, max_grad_norm=1.0,
                  expected_batch_size=wl.B, noise_seed=3)
opt.attach_grad_sample_module(gsm)
r = bench.hot_path(_native, gsm, opt, wl, [(0, wl.B)], steps=iters)
print({k: r[k] for k in ("batch", "ms", "samples_per_s", "hbm_GBs", "hbm_frac", "mfma_tflops")}, "P_arena", gsm.arena.P)
