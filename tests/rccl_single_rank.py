"""Helper of test_rccl_gpu.py (run as a subprocess so the process group dies with it): a one-rank RCCL group on
cuda:0 — the only RCCL configuration a one-GPU box can run — under DistributedDPOptimizer + GraphedDPStep.
Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dp_timeautodiff_amd import _native, models  # noqa: E402
from dp_timeautodiff_amd.distributed import DifferentiallyPrivateDistributedDataParallel as DPDDP  # noqa: E402
from dp_timeautodiff_amd.grad_sample import GradSampleModule  # noqa: E402
from dp_timeautodiff_amd.graphed import GraphedDPStep  # noqa: E402
from dp_timeautodiff_amd.layers.fused_gru import fuse_recurrences  # noqa: E402
from dp_timeautodiff_amd.optimizer import DistributedDPOptimizer, DPOptimizer  # noqa: E402


def run(distributed: bool, steps: int):
    torch.manual_seed(0)
    B, T, K, H = 48, 8, 6, 32
    m = fuse_recurrences(models.DeapStackDP(64, B, T, 0, 0, K, [], K, H, 1, 16, 8, 3).cuda().train())
    if distributed:
        DPDDP(m)
    gsm = GradSampleModule(m)
    adam = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    cls = DistributedDPOptimizer if distributed else DPOptimizer
    opt = cls(adam, noise_multiplier=1.0, max_grad_norm=0.05, expected_batch_size=B, noise_seed=17)
    opt.attach_grad_sample_module(gsm)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, T, K, generator=g).cuda()
    eps = torch.randn(B, T, 3, generator=g).cuda()

    def loss_fn():
        m.injected_eps = eps
        return models.vae_loss(gsm, x)

    hooks = []
    opt.attach_step_hook(lambda o: hooks.append(1))
    gr = GraphedDPStep(gsm, opt, loss_fn, warmup=2).capture()
    for _ in range(steps):
        loss = gr()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    return flat, float(loss), gr, len(hooks), opt.noise_state()["offset"]


def main():
    _native.load()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[1]}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)  # RCCL communicator up before anything is captured
    single, loss_s, _, hooks_s, off_s = run(False, 3)
    multi, loss_m, gr, hooks_m, off_m = run(True, 3)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(),
           "rccl": ".".join(str(v) for v in torch.cuda.nccl.version()),
           "exchange_captured": bool(gr.exchange_captured), "distributed": bool(gr.distributed),
           "max_abs_diff": float((single - multi).abs().max()), "max_abs": float(single.abs().max()),
           "loss_single": loss_s, "loss_multi": loss_m, "hooks": [hooks_s, hooks_m], "offsets": [off_s, off_m]}
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
