#!/usr/bin/env python3
"""bench.py — DP-SGD training samples/s of the MI355X-native step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one complete DP-SGD training step on a resident synthetic batch: zero_grad, forward, loss,
backward (hooks -> per-sample-gradient kernels K1/K3/K4), optimizer.step (K6 clip factors, K7 clipped sum +
Philox noise + scale, then Adam).  Workload at N=1 = BASELINE.json configs[1] (VAE, B=1024, T=24, 32 features,
hidden 128); with N>1 every rank runs the same per-GPU batch (weak scaling), clipped sums are all-reduced as
one flat buffer over RCCL and identical Philox noise is added on every rank.

Printed JSON (rank 0, one line) also carries
  roofline      the dominant kernel (K7 clip_reduce_noise): algorithmic bytes 4*P*B per launch / mean launch
                duration measured with HIP events on the launch stream inside the timed region;
  hot_path      K1..K7 alone, replayed from captured activations/backprops (no autograd, no Adam);
  cpu_baseline  the oracle's torch-CPU restatement of the reference step (per-call einsum + accumulate,
                two-pass clip), timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--physical-batch", type=int, default=None, help="max physical chunk (BatchMemoryManager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="leave the model's library GEMMs on hipBLASLt's default heuristics (no TunableOp)")
    ap.add_argument("--hooked-gru", action="store_true",
                    help="keep the reference's T-step GRU loop + per-call hooks instead of the fused K8 recurrence")
    return ap.parse_args()


class KernelTimer:
    """Wraps _native entry points (one kernel family) with HIP events recorded on the launch stream.  Wrapped entry
    points may call each other (the single-layer K1 call forwards strided operands to the grouped one): only the
    outermost call is timed."""

    def __init__(self, native, *names):
        self.native, self.names = native, names
        self.events = []
        self.enabled = False
        self.depth = 0
        for name in names:
            setattr(native, name, self._wrap(getattr(native, name)))

    def _wrap(self, orig):
        def call(*a, **k):
            if not self.enabled or self.depth:
                return orig(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()  # torch's current stream == the stream _native launches on
            self.depth += 1
            try:
                r = orig(*a, **k)
            finally:
                self.depth -= 1
            e1.record()
            self.events.append((e0, e1))
            return r
        return call

    def mean_ms(self):
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events)


def cpu_baseline(cfg_name, sample_B, budget_s=25.0):
    """Reference op sequence on host cores (oracle = CPU restatement of the Opacus step)."""
    from dp_timeautodiff_amd import workloads
    from oracle import opacus_ref as O
    cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    torch.set_num_threads(cores)
    wl = workloads.build(cfg_name, torch.device("cpu"), batch=sample_B)
    gsm = O.OracleGradSampleModule(wl.model)
    opt = O.OracleDPOptimizer(list(wl.model.parameters()), noise_multiplier=1.0, max_grad_norm=1.0,
                              expected_batch_size=sample_B, generator=torch.Generator().manual_seed(3))
    adam = torch.optim.Adam(wl.model.parameters(), lr=2e-4, weight_decay=1e-6)

    def step():
        opt.zero_grad()
        wl.loss_fn(wl.model).backward()
        opt.pre_step()
        adam.step()

    t0 = time.perf_counter()
    step()  # warm-up
    first = time.perf_counter() - t0
    n = max(1, min(5, int(budget_s / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    return {"value": sample_B / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} step(s) of {cfg_name} at batch {sample_B} (torch-CPU restatement of the Opacus step: "
                      f"per-call einsum+accumulate, two-pass clip, per-parameter noise, Adam), {dt*1e3:.0f} ms/step"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal of the N>1 code path on a one-GPU box: every rank on cuda:0, collectives through gloo
    rehearsal = os.environ.get("DPK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(75, exit=True)  # a hang names its line instead of being killed silently
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from dp_timeautodiff_amd import _native, workloads
    from dp_timeautodiff_amd.distributed import DifferentiallyPrivateDistributedDataParallel as DPDDP
    from dp_timeautodiff_amd.grad_sample import GradSampleModule
    from dp_timeautodiff_amd.optimizer import DistributedDPOptimizer, DPOptimizer
    _native.load()
    tuned = False
    if not args.no_gemm_tuning and args.config != "cfg5":
        # hipBLASLt/rocBLAS solution per GEMM shape, chosen during the warm-up steps (seconds for cfg1-3; cfg5's
        # 2048-wide GEMMs over 524k rows take ~6 ms per candidate run, i.e. minutes to tune, and are already at the
        # library's peak: left on the default heuristics)
        from dp_timeautodiff_amd import gemm_tuning
        try:
            tuned = gemm_tuning.enable()
        except Exception as e:  # an optional speed-up of the model's own GEMMs: never a reason to fail the bench
            print(f"[bench] GEMM tuning unavailable: {e}", file=sys.stderr)

    wl = workloads.build(args.config, dev, batch=args.batch, seed_offset=rank * 1000,
                         fuse_recurrence=not args.hooked_gru)
    B = wl.B
    phys = args.physical_batch or (2048 if args.config == "cfg5" else B)  # cfg5: 49 MB/sample -> 100 GB arena
    model = wl.model
    if world > 1:
        DPDDP(model)  # broadcasts rank-0 parameters
    gsm = GradSampleModule(model)
    adam = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-6)
    cls = DistributedDPOptimizer if world > 1 else DPOptimizer
    opt = cls(adam, noise_multiplier=1.0, max_grad_norm=1.0, expected_batch_size=B, noise_seed=3)
    opt.attach_grad_sample_module(gsm)
    if rehearsal and world > 1:
        # gloo's own CUDA staging hung on this pool's one-GPU boxes; stage through the host explicitly
        class _HostStaged:
            ReduceOp = dist.ReduceOp

            @staticmethod
            def all_reduce(t, op=None, group=None):
                h = t.detach().cpu()
                dist.all_reduce(h, op=op, group=group)
                t.copy_(h)

            def __getattr__(self, name):
                return getattr(dist, name)
        opt._dist = _HostStaged()
    chunks = [(lo, min(lo + phys, B)) for lo in range(0, B, phys)]

    k7 = KernelTimer(_native, "clip_reduce_noise" if world == 1 else "clip_reduce")
    k1 = KernelTimer(_native, "linear_grad_sample", "linear_grad_sample_group")

    graphed = None
    if not args.eager and len(chunks) == 1:
        from dp_timeautodiff_amd.graphed import GraphedDPStep
        graphed = GraphedDPStep(gsm, opt, lambda: wl.loss_fn(gsm, 0, B), warmup=2).capture()

    def eager_step():
        for ci, (lo, hi) in enumerate(chunks):
            opt.signal_skip_step(do_skip=ci < len(chunks) - 1)
            opt.zero_grad()
            loss = wl.loss_fn(gsm, lo, hi)
            loss.backward()
            opt.step()
        return loss

    def step():
        return graphed() if graphed is not None else eager_step()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    k7.enabled = k1.enabled = graphed is None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    n_timed = args.steps
    if graphed is not None:
        # a graph replay issues no per-kernel host calls: time the kernels (HIP events on the launch stream) over
        # eager re-runs of the very same step, right after the timed region
        k7.enabled = k1.enabled = True
        n_timed = min(args.steps, 5)
        for _ in range(n_timed):
            eager_step()
        sync()
    k7.enabled = k1.enabled = False
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss)

    # ---- roofline of the dominant kernel (K7) ------------------------------------------------------------
    P_arena = gsm.arena.P
    k7_ms = k7.mean_ms()
    rows_per_launch = B / len(chunks)
    k7_bytes = 4.0 * P_arena * rows_per_launch  # each per-sample gradient element read once
    roof = {"bound": "hbm", "kernel": "clip_reduce_kernel (K7: clipped sum + noise + scale)",
            "achieved": k7_bytes / (k7_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": k7_bytes / (k7_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
            "avg_launch_ms": k7_ms, "bytes_per_launch": k7_bytes}
    traffic, traffic_src = pmc_traffic(args.config, B, len(chunks))
    roof["traffic"], roof["traffic_source"] = traffic, traffic_src
    k1_total_ms = k1.total_ms() / n_timed
    k1_bytes = B * (4.0 * wl.n_params + wl.io_bytes_per_sample)
    k1_roof = {"bound": "hbm", "kernel": "linear_gs_kernel (K1, all Linear layers of one step)",
               "achieved": k1_bytes / (k1_total_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": k1_bytes / (k1_total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_step": k1_total_ms,
               "tflops": B * wl.flops_per_sample / (k1_total_ms * 1e-3) / 1e12,
               "mfma_frac": B * wl.flops_per_sample / (k1_total_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS}

    # ---- hot path alone: K1..K7 replayed from captured (activation, backprop) pairs -------------------------
    hot = None
    if world == 1:
        hot = hot_path(_native, gsm, opt, wl, chunks, steps=max(args.steps, 5))

    ms = dt / args.steps * 1e3
    out = {
        "metric": "DP-SGD training samples/sec (whole node)", "value": world * B * args.steps / dt,
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {workloads.CONFIGS[args.config]['desc']}",
                   "per_gpu_batch": B, "global_batch": B * world, "physical_batch": phys,
                   "params": wl.n_params, "parallelism": f"dp{world}",
                   "step": "zero_grad+forward(K8)+backward(K8; per-sample grads K1/K3/K4)+DPOptimizer.step(K6,K7)+Adam(K9)",
                   "gru": "hooked T-step loop" if args.hooked_gru else "fused K8 recurrence",
                   "library_gemms": "TunableOp (tuned in warm-up)" if tuned else "default heuristics",
                   "launch": ("hipGraph replay of the captured step" + (" (local part; all-reduce+noise+Adam eager)"
                              if world > 1 else "")) if graphed is not None else "eager"},
        "roofline": roof, "roofline_k1": k1_roof, "hot_path": hot, "final_loss": final_loss,
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            sample = args.cpu_sample or {"cfg2": 256, "cfg1": 64, "cfg3": 64, "cfg5": 8}.get(args.config, 64)
            out["cpu_baseline"] = cpu_baseline(args.config, min(sample, B))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(config, B, n_chunks):
    """HBM bytes per K7 launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, collected in
    separate runs by tools/hotpath_replay.py + tools/pmc_summary.py at this config's default batch)."""
    import glob
    from dp_timeautodiff_amd import workloads
    if n_chunks != 1 or B != workloads.CONFIGS[config]["B"]:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_pmc_traffic.json")))
    if not files:
        return None, None
    k = json.load(open(files[-1]))["kernels"]
    tot = sum(v["fetch_bytes"] + v["write_bytes"] for n, v in k.items()
              if n.startswith("clip_reduce_kernel") or n.startswith("finalize_kernel"))
    return tot, os.path.relpath(files[-1], ROOT)


def hot_path(native, gsm, opt, wl, chunks, steps):
    """Time K1..K7 alone on resident (A, G) pairs captured from one real step of the first chunk."""
    captured = []
    orig = {n: getattr(native, n) for n in ("linear_grad_sample", "linear_grad_sample_group", "embedding_grad_sample",
                                            "layernorm_grad_sample")}
    depth = [0]

    def cap(name):
        def f(*a, **k):
            if depth[0] == 0:  # the single-layer K1 call may forward to the grouped one: record the outermost only
                captured.append((name, a, k))
            depth[0] += 1
            try:
                return orig[name](*a, **k)
            finally:
                depth[0] -= 1
        return f

    for n in orig:
        setattr(native, n, cap(n))
    lo, hi = chunks[0]
    opt.signal_skip_step(do_skip=False)
    opt.zero_grad()
    wl.loss_fn(gsm, lo, hi).backward()
    for n in orig:
        setattr(native, n, orig[n])
    arena = gsm.arena
    Bc = hi - lo
    inv = 1.0 / wl.B
    ws = arena.ensure_workspace(Bc)
    full = native.FlatView(arena.buf, 0, arena.P)

    def replay():
        for name, a, k in captured:
            orig[name](*a, **k)
        native.clip_factors(arena.part, arena.n_slots, arena.clip, arena.norms, Bc, 1.0)
        native.clip_reduce_noise(full, arena.clip, None, None, 3, 0, 1.0, inv, arena.summed, arena.grad, Bc, ws)

    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    opt.zero_grad()
    bytes_ps = 8.0 * wl.n_params + wl.io_bytes_per_sample
    sps = Bc / (ms * 1e-3)
    return {"what": "K1..K7 only (per-sample grads from resident activations/backprops -> norms -> clip -> sum -> "
                    "noise -> scale), one physical chunk", "batch": Bc, "ms": ms, "samples_per_s": sps,
            "bytes_per_sample": bytes_ps, "hbm_GBs": sps * bytes_ps / 1e9, "hbm_frac": sps * bytes_ps / 1e9 / HBM_PEAK_GBS,
            "mfma_tflops": sps * wl.flops_per_sample / 1e12,
            "mfma_frac": sps * wl.flops_per_sample / 1e12 / MFMA_F32_PEAK_TFLOPS}


if __name__ == "__main__":
    main()
