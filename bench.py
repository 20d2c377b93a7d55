#!/usr/bin/env python3
"""bench.py — DP-SGD training samples/s of the MI355X-native step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): this process touches no GPU and starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` on itself, one rank per
GPU over RCCL; under a launcher (the driver's torch.distributed.run) it is one of the ranks.  Either way WORLD_SIZE
must equal --gpus or the run fails: a line that says n_gpus: N was measured on N ranks.

A "step" is one complete DP-SGD training step on a resident synthetic batch: zero_grad, forward, loss, backward
(hooks -> per-sample-gradient kernels K1/K3/K4), optimizer.step (K6 clip factors, K7 clipped sum + Philox noise +
scale, then Adam K9).  Workload at N=1 = BASELINE.json configs[2], the LARGEST single-GPU configuration (latent
diffusion denoiser, B=4096, T=48, per-sample clip over all 32 parameter groups; `--config cfg2` = configs[1], the VAE at
B=1024, is measured in the same run under other_configs); with N>1 every rank runs the same per-GPU batch (weak
scaling), the flat clipped sums are all-reduced once per step over RCCL and identical Philox noise is added on every
rank.  `--config cfg4` = configs[3]: the VAE and the latent denoiser, 2048 samples per GPU through both.

Printed JSON (rank 0, one line) also carries
  roofline       the kernel that is dominant BY TIME, K1 (per-sample-gradient contraction, all Linear layers of a
                 step): algorithmic bytes of its launches / their summed duration, HIP events on the launch stream;
  roofline_k7    the same for K7 (clip_reduce_noise, one pass over the arena);
  hot_path       K1..K7 alone, replayed from captured activations/backprops (no autograd, no Adam);
  other_configs  (N=1, default config only) the same measurements at cfg2 (configs[1]: VAE B=1024) and cfg1 (the
                 reference's own dp_job.sh shape), with the headline's --steps / --warmup;
  dist           (N>1) backend, world size and RCCL version as torch.distributed reports them;
  cpu_baseline   the oracle's torch-CPU restatement of the reference step (per-call einsum + accumulate,
                 two-pass clip), timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2516.6  # dense; K1 at T > 32 forms each fp32 product from 6 bf16 partial products on this pipe
K1_SPLIT_TERMS = 6


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--physical-batch", type=int, default=None, help="max physical chunk (BatchMemoryManager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the cfg2 / cfg1 legs of the default run")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--cpu-sample", type=int, default=None)
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="leave the model's library GEMMs on hipBLASLt's default heuristics (no TunableOp)")
    ap.add_argument("--hooked-gru", action="store_true",
                    help="keep the reference's T-step GRU loop + per-call hooks instead of the fused K8 recurrence")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check only: start the ranks, form the process group (gloo, CPU), all-reduce one "
                         "word, print n_gpus / dist — no GPU is touched")
    return ap.parse_args(argv)


# ---- launching N ranks ------------------------------------------------------------------------------------------

def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, gpus: int, port: int):
    """The command this process runs to become N ranks (same flags, one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def needs_spawn(args, environ) -> bool:
    return args.gpus > 1 and "WORLD_SIZE" not in environ


def spawn_ranks(args, argv) -> int:
    """Parent of a `--gpus N` run without a launcher.  Nothing here may initialise the GPU: the ranks are children."""
    cmd = launch_command(argv, args.gpus, free_port())
    env = dict(os.environ, DPK_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dist_info(dist):
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    if info["backend"] == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # reported, never fatal
            info["rccl_version"] = f"unavailable ({e})"
    return info


def dry_run(args, rank, world):
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
    else:
        seen = 1
    if seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the all-reduce saw {seen} ranks")
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": seen, "config": {"workload": args.config},
                          "dist": dist_info(dist) if world > 1 else None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---- kernel timing ----------------------------------------------------------------------------------------------

class KernelTimer:
    """Brackets the C-ABI entry points of one kernel family with HIP events recorded on the launch stream.  The ctypes
    functions of the loaded library are wrapped, not the Python wrappers around them: the host-side argument
    marshalling of a call must not sit between the two records (with an idle GPU the first event fires at once and
    the interval would include it — for the small-batch configs that was most of the reading)."""

    def __init__(self, native, *symbols):
        self.native, self.symbols = native, symbols
        self.events = []
        self.enabled = False
        lib = native.load()
        for name in symbols:
            setattr(lib, name, self._wrap(getattr(lib, name)))  # instance attribute shadows the ctypes function

    def _wrap(self, orig):
        def call(*a):
            if not self.enabled:
                return orig(*a)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()  # torch's current stream == the stream _native launches on
            r = orig(*a)
            e1.record()
            self.events.append((e0, e1))
            return r
        return call

    def reset(self):
        self.events = []

    def mean_ms(self):
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)

    def total_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events)


_TIMERS = {}


def timers(native):
    if not _TIMERS:
        _TIMERS["k7"] = KernelTimer(native, "dpk_clip_reduce_noise", "dpk_clip_reduce")
        _TIMERS["k1"] = KernelTimer(native, "dpk_linear_grad_sample", "dpk_linear_grad_sample_group",
                                    "dpk_linear_grad_sample_multi")
    return _TIMERS["k1"], _TIMERS["k7"]


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
    del gsm
    return {"value": sample_B / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} step(s) of {cfg_name} at batch {sample_B} (torch-CPU restatement of the Opacus step: "
                      f"per-call einsum+accumulate, two-pass clip, per-parameter noise, Adam), {dt*1e3:.0f} ms/step"}


# ---- one model under DP-SGD -------------------------------------------------------------------------------------

class Part:
    """One model of a config wrapped for DP-SGD (cfg4 has two: the VAE and the denoiser)."""

    def __init__(self, name, ctx, args, batch=None):
        from dp_timeautodiff_amd import workloads
        from dp_timeautodiff_amd.distributed import DifferentiallyPrivateDistributedDataParallel as DPDDP
        from dp_timeautodiff_amd.grad_sample import GradSampleModule
        from dp_timeautodiff_amd.optimizer import DistributedDPOptimizer, DPOptimizer
        self.name, self.ctx = name, ctx
        dist, world = ctx["dist"], ctx["world"]
        self.wl = wl = workloads.build(name, ctx["dev"], batch=batch, seed_offset=ctx["rank"] * 1000,
                                       fuse_recurrence=not args.hooked_gru)
        self.B = B = wl.B
        self.phys = args.physical_batch or workloads.CONFIGS[name].get("phys", B)
        if world > 1:
            DPDDP(wl.model)  # broadcasts rank-0 parameters
        self.gsm = GradSampleModule(wl.model)
        adam = torch.optim.Adam(wl.model.parameters(), lr=2e-4, weight_decay=1e-6)
        cls = DistributedDPOptimizer if world > 1 else DPOptimizer
        self.opt = cls(adam, noise_multiplier=1.0, max_grad_norm=1.0, expected_batch_size=B, noise_seed=3)
        self.opt.attach_grad_sample_module(self.gsm)
        if ctx["rehearsal"] and world > 1:
            # one-GPU rehearsal of the N>1 code path: gloo's own CUDA staging hung on this pool's one-GPU boxes;
            # stage through the host explicitly
            class _HostStaged:
                ReduceOp = dist.ReduceOp

                @staticmethod
                def all_reduce(t, op=None, group=None):
                    h = t.detach().cpu()
                    dist.all_reduce(h, op=op, group=group)
                    t.copy_(h)

                def __getattr__(self, item):
                    return getattr(dist, item)
            self.opt._dist = _HostStaged()
        self.chunks = [(lo, min(lo + self.phys, B)) for lo in range(0, B, self.phys)]
        self.graphed = None
        if not args.eager and len(self.chunks) == 1:
            from dp_timeautodiff_amd.graphed import GraphedDPStep
            try:
                self.graphed = GraphedDPStep(self.gsm, self.opt, lambda: wl.loss_fn(self.gsm, 0, B), warmup=2).capture()
            except Exception as e:  # noqa: BLE001
                if world == 1:
                    raise
                # a multi-rank capture that the collective library refuses fails on every rank alike: say so and time
                # eager steps instead of printing nothing (the line's config.launch then reads "eager")
                print(f"[bench] rank {ctx['rank']}: graph capture of the data-parallel step failed ({type(e).__name__}: "
                      f"{e}); falling back to eager launches", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                self.graphed = None

    def eager_step(self):
        opt, wl = self.opt, self.wl
        for ci, (lo, hi) in enumerate(self.chunks):
            opt.signal_skip_step(do_skip=ci < len(self.chunks) - 1)
            opt.zero_grad()
            loss = wl.loss_fn(self.gsm, lo, hi)
            loss.backward()
            opt.step()
        return loss

    def step(self):
        return self.graphed() if self.graphed is not None else self.eager_step()

    def launch_mode(self):
        if self.graphed is None:
            return "eager"
        if self.graphed.distributed and not self.graphed.exchange_captured:
            return "hipGraph replay of the local part; all-reduce + noise + Adam eager"
        if self.graphed.distributed:
            return "hipGraph replay of the whole step (RCCL all-reduce, noise and Adam are graph nodes)"
        return "hipGraph replay of the captured step"


def run_config(cfg, ctx, args, steps, warmup, batch=None, want_hot=True):
    """Build the config's model(s), time `steps` steps after `warmup`, then the kernel legs.  Returns the result dict."""
    from dp_timeautodiff_amd import _native, workloads
    dist, world, dev = ctx["dist"], ctx["world"], ctx["dev"]
    k1, k7 = timers(_native)
    tuned = False
    if not args.no_gemm_tuning and not workloads.CONFIGS[cfg].get("no_gemm_tuning"):
        # hipBLASLt/rocBLAS solution per GEMM shape, chosen during the warm-up steps (seconds for cfg1-4; cfg5's
        # 2048-wide GEMMs over 524k rows take ~6 ms per candidate run, i.e. minutes to tune, and are already at the
        # library's peak: left on the default heuristics)
        from dp_timeautodiff_amd import gemm_tuning
        try:
            tuned = gemm_tuning.enable()
        except Exception as e:  # an optional speed-up of the model's own GEMMs: never a reason to fail the bench
            print(f"[bench] GEMM tuning unavailable: {e}", file=sys.stderr)
    names = workloads.CONFIGS[cfg].get("parts", (cfg,))
    parts = [Part(n, ctx, args, batch=batch) for n in names]
    B = parts[0].B
    if any(p.B != B for p in parts):
        raise SystemExit(f"{cfg}: parts disagree on the per-GPU batch")

    def step():
        for p in parts:
            loss = p.step()
        return loss

    def eager_step():
        for p in parts:
            p.eager_step()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    graphed = all(p.graphed is not None for p in parts)
    for _ in range(warmup):
        step()
    sync()
    k1.reset(); k7.reset()
    k7.enabled = k1.enabled = not graphed
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    n_timed = steps
    if graphed:
        # a graph replay issues no per-kernel host calls: time the kernels (HIP events on the launch stream) over
        # eager re-runs of the very same step, right after the timed region
        k7.enabled = k1.enabled = True
        n_timed = min(steps, 5)
        for _ in range(n_timed):
            eager_step()
        sync()
    k7.enabled = k1.enabled = False
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if ctx["rehearsal"]:
            h = tmax.cpu(); dist.all_reduce(h, op=dist.ReduceOp.MAX); tmax = h
        else:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss)

    # ---- rooflines: K1 (dominant by time) and K7 ------------------------------------------------------------
    n_params = sum(p.wl.n_params for p in parts)
    io_bytes = sum(p.wl.io_bytes_per_sample for p in parts)
    flops = sum(p.wl.flops_per_sample for p in parts)
    k1_total_ms = k1.total_ms() / n_timed
    k1_launches = len(k1.events) / n_timed
    k1_bytes = B * (4.0 * n_params + io_bytes)  # per step: every gradient element written once + operand reads
    k1_gbs = k1_bytes / (k1_total_ms * 1e-3) / 1e9
    k1_tf = B * flops / (k1_total_ms * 1e-3) / 1e12
    # K1's large-layer kernel: fp32 MFMA up to 32 steps; beyond, fp32 products as K1_SPLIT_TERMS exact bf16 partial
    # products (fp32 accumulate) on the bf16 pipe, whose peak in ALGORITHMIC fp32 flops is the bf16 peak / terms
    T_seq = max(workloads.CONFIGS[p.name]["T"] for p in parts)
    split = T_seq > 32 and os.environ.get("DPK_K1_MFMA", "") != "f32"
    role_split = split and T_seq <= 48 and os.environ.get("DPK_K1_WS", "") != "0"
    long_seq = split and T_seq > 48 and os.environ.get("DPK_K1_WS", "") != "0"
    mfma_peak = MFMA_BF16_PEAK_TFLOPS / K1_SPLIT_TERMS if split else MFMA_F32_PEAK_TFLOPS
    ridge = mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    ai = B * flops / k1_bytes
    traffic = pmc_traffic(cfg, B, parts)
    roof = {"bound": "hbm" if ai < ridge else "mfma",
            "kernel": ("linear_gs_ws_kernel (role-split, 32 < T <= 48)" if role_split else
                       "linear_gs_wsl_kernel (T > 48: 256 x 256 super-tiles)" if long_seq else
                       "linear_gs_split_kernel" if split else "linear_gs_persist_wide_kernel") + " / linear_gs_kernel "
                      "(K1: per-sample-gradient contraction, every Linear layer of one step)",
            "arithmetic": (f"fp32 in/out, fp32 accumulate; products = {K1_SPLIT_TERMS} of the 9 exact partial products of a "
                           "3-way bf16 split (v_mfma_f32_32x32x16_bf16); dropped < 2^-26 of a product" if split else
                           "fp32 (v_mfma_f32_32x32x2_f32)"),
            "achieved": k1_gbs if ai < ridge else k1_tf, "peak": HBM_PEAK_GBS if ai < ridge else mfma_peak,
            "unit": "GB/s" if ai < ridge else "TFLOP/s",
            "frac": (k1_gbs / HBM_PEAK_GBS) if ai < ridge else (k1_tf / mfma_peak),
            # `traffic` would be HBM bytes per step from PMC counters of THIS run; bench.py cannot collect them (that
            # takes rocprofv3 --pmc passes of their own), so it is null and the figure of the committed passes over
            # tools/hotpath_replay.py at this config is carried under its own name
            "traffic": None, "traffic_committed": traffic.get("k1"), "traffic_source": traffic.get("source"),
            "ms_per_step": k1_total_ms, "launches_per_step": k1_launches,
            "avg_launch_ms": k1_total_ms / max(k1_launches, 1), "bytes_per_step": k1_bytes,
            "hbm_GBs": k1_gbs, "hbm_frac": k1_gbs / HBM_PEAK_GBS, "tflops": k1_tf,
            "mfma_frac": k1_tf / mfma_peak, "mfma_peak_tflops": mfma_peak, "flop_per_byte": ai}
    k7_ms = k7.mean_ms()
    k7_launches = len(k7.events) / n_timed
    # algorithmic: every per-sample gradient element read once, 4 * P * B (the arena's row padding, ~2 % more at cfg3,
    # is traffic the kernel really moves but not work the step asked for)
    k7_bytes = sum(4.0 * p.wl.n_params * p.B for p in parts) / max(k7_launches, 1)
    roof7 = {"bound": "hbm", "kernel": "clip_reduce_kernel (K7: clipped sum + noise + scale)",
             "achieved": k7_bytes / (k7_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": k7_bytes / (k7_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
             "traffic_committed": traffic.get("k7"), "traffic_source": traffic.get("source"),
             "arena_bytes_per_launch": sum(4.0 * p.gsm.arena.P * p.B for p in parts) / max(k7_launches, 1),
             "avg_launch_ms": k7_ms, "bytes_per_launch": k7_bytes,
             "launches_per_step": k7_launches}

    # ---- hot path alone: K1..K7 replayed from captured (activation, backprop) pairs ---------------------------
    hot = None
    if world == 1 and want_hot:
        hots = [hot_path(_native, p.gsm, p.opt, p.wl, p.chunks, steps=max(steps, 5)) for p in parts]
        hot = hots[0] if len(hots) == 1 else combine_hot(hots)

    ms = dt / steps * 1e3
    out = {
        "ms_per_step": ms, "samples_per_s": world * B * steps / dt, "final_loss": final_loss,
        "config": {"workload": f"{cfg}: {workloads.CONFIGS[cfg]['desc']}",
                   "per_gpu_batch": B, "global_batch": B * world, "physical_batch": parts[0].phys,
                   "params": n_params, "parallelism": f"dp{world}",
                   "step": "zero_grad+forward(K8/K10)+backward(K8/K10; per-sample grads K1/K3/K4)+DPOptimizer.step(K6,K7)"
                           "+Adam(K9)" + (", once per model" if len(parts) > 1 else ""),
                   "gru": "hooked T-step loop" if args.hooked_gru else "fused recurrence kernels",
                   "library_gemms": "TunableOp (tuned in warm-up)" if tuned else "default heuristics",
                   "launch": parts[0].launch_mode()},
        "roofline": roof, "roofline_k7": roof7, "hot_path": hot,
    }
    for p in parts:
        p.gsm._close()
    del parts
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def combine_hot(hots):
    ms = sum(h["ms"] for h in hots)
    Bc = hots[0]["batch"]
    bytes_ps = sum(h["bytes_per_sample"] for h in hots)
    flops_ps = sum(h["mfma_tflops"] * 1e12 / h["samples_per_s"] for h in hots)
    sps = Bc / (ms * 1e-3)
    return {"what": hots[0]["what"] + " — both models", "batch": Bc, "ms": ms, "samples_per_s": sps,
            "bytes_per_sample": bytes_ps, "hbm_GBs": sps * bytes_ps / 1e9, "hbm_frac": sps * bytes_ps / 1e9 / HBM_PEAK_GBS,
            "mfma_tflops": sps * flops_ps / 1e12, "mfma_frac": sps * flops_ps / 1e12 / MFMA_F32_PEAK_TFLOPS}


def summary_of(r):
    """The short form of a run_config result kept under other_configs."""
    h, k = r["hot_path"] or {}, r["roofline"]
    return {"workload": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "samples_per_s": r["samples_per_s"],
            "launch": r["config"]["launch"],
            "hot_path": {"ms": h.get("ms"), "hbm_frac": h.get("hbm_frac"), "mfma_frac": h.get("mfma_frac")},
            "roofline_k1": {"bound": k["bound"], "frac": k["frac"], "hbm_frac": k["hbm_frac"],
                            "mfma_frac": k["mfma_frac"], "ms_per_step": k["ms_per_step"]},
            "roofline_k7": {"frac": r["roofline_k7"]["frac"], "avg_launch_ms": r["roofline_k7"]["avg_launch_ms"]}}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if needs_spawn(args, os.environ):
        raise SystemExit(spawn_ranks(args, argv))  # no GPU call has happened in this process
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for the wrong rank count")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal of the N>1 code path on a one-GPU box: every rank on cuda:0, collectives through gloo
    rehearsal = os.environ.get("DPK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)  # a hang names its line instead of being killed silently
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from dp_timeautodiff_amd import _native, workloads
    _native.load()
    ctx = {"dev": dev, "rank": rank, "world": world, "dist": dist, "rehearsal": rehearsal}
    if args.config not in workloads.CONFIGS:
        raise SystemExit(f"unknown --config {args.config}; known: {sorted(workloads.CONFIGS)}")

    r = run_config(args.config, ctx, args, args.steps, args.warmup, batch=args.batch)
    out = {
        "metric": "DP-SGD training samples/sec (whole node)", "value": r["samples_per_s"],
        "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "config": r["config"], "roofline": r["roofline"],
        "roofline_k7": r["roofline_k7"], "hot_path": r["hot_path"], "final_loss": r["final_loss"],
    }
    if dist is not None:
        out["dist"] = dist_info(dist)
    if world == 1 and args.config == "cfg3" and args.batch is None and not args.no_other_configs:
        # configs[1] (the VAE at B=1024) and the reference's literal dp_job.sh shape (cfg1), same steps / warm-up
        out["other_configs"] = {}
        for name, (k, w) in (("cfg2", (args.steps, args.warmup)), ("cfg1", (max(args.steps, 20), args.warmup))):
            try:
                out["other_configs"][name] = summary_of(run_config(name, ctx, args, k, w))
            except Exception as e:  # the headline line must survive a failure of an extra leg; the failure is reported
                out["other_configs"][name] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            sample = args.cpu_sample or workloads.CONFIGS[args.config].get("cpu_sample", 64)
            base_cfg = workloads.CONFIGS[args.config].get("parts", (args.config,))[0]
            out["cpu_baseline"] = cpu_baseline(base_cfg, min(sample, out["config"]["per_gpu_batch"]))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(config, B, parts):
    """HBM bytes per step of K1 and per launch of K7 from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 +
    WRITE_SIZE, collected in separate runs by tools/hotpath_replay.py + tools/pmc_summary.py at this config's default
    batch and one physical chunk)."""
    import glob
    from dp_timeautodiff_amd import workloads
    if len(parts) != 1 or len(parts[0].chunks) != 1 or B != workloads.CONFIGS[config]["B"]:
        return {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_pmc_traffic.json")))
    if not files:
        return {}
    doc = json.load(open(files[-1]))
    k = doc["kernels"]
    tot7 = sum(v["fetch_bytes"] + v["write_bytes"] for n, v in k.items()
               if n.startswith("clip_reduce_kernel") or n.startswith("finalize_kernel"))
    return {"k7": tot7, "k1": doc.get("k1_bytes_per_step"), "source": os.path.relpath(files[-1], ROOT)}


def hot_path(native, gsm, opt, wl, chunks, steps):
    """Time K1..K7 alone on resident (A, G) pairs captured from one real step of the first chunk."""
    captured = []
    names = ("linear_grad_sample", "linear_grad_sample_group", "linear_grad_sample_multi", "embedding_grad_sample",
             "layernorm_grad_sample")
    orig = {n: getattr(native, n) for n in names if hasattr(native, n)}
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
    opt._step_skip_queue.clear()
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
    # the launches of one pass as a hipGraph: at the small-batch configs the host cannot issue a dozen launches in the
    # 0.1 ms they take, and the reading would be the host's
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        replay()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    opt.zero_grad()
    bytes_ps = 8.0 * wl.n_params + wl.io_bytes_per_sample
    sps = Bc / (ms * 1e-3)
    return {"what": "K1..K7 only (per-sample grads from resident activations/backprops -> norms -> clip -> sum -> "
                    "noise -> scale), one physical chunk, replayed as a hipGraph", "batch": Bc, "ms": ms, "samples_per_s": sps,
            "bytes_per_sample": bytes_ps, "hbm_GBs": sps * bytes_ps / 1e9, "hbm_frac": sps * bytes_ps / 1e9 / HBM_PEAK_GBS,
            "mfma_tflops": sps * wl.flops_per_sample / 1e12,
            "mfma_frac": sps * wl.flops_per_sample / 1e12 / MFMA_F32_PEAK_TFLOPS}


if __name__ == "__main__":
    main()
