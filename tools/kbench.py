"""Micro-benchmark of the individual kernels at the BASELINE.json config shapes (GPU box)."""
import sys, os, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dp_timeautodiff_amd import _native as nat

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def bench_linear(B, T, in_f, out_f):
    A = torch.randn(B, T, in_f, device="cuda"); G = torch.randn(B, T, out_f, device="cuda")
    ld = (in_f + 31) // 32 * 32 if (in_f >= 64 and in_f % 32) else in_f
    n = ld * out_f
    stride = (n + 63) // 64 * 64 + 64
    base = torch.zeros(B, stride, device="cuda")
    part = torch.zeros(B, nat.linear_slots(in_f, out_f), device="cuda")
    vw = nat.FlatView(base, 0, n); vb = nat.FlatView(base, (n + 63) // 64 * 64, out_f) if out_f <= 64 else None
    ms = timeit(lambda: nat.linear_grad_sample(A, G, vw, vb, part, 0, float(B), ld_w=ld))
    wbytes = B * (in_f * out_f) * 4; rbytes = B * T * (in_f + out_f) * 4
    flops = 2.0 * B * T * n
    print(f"K1 B={B} T={T} in={in_f} out={out_f}: {ms*1e3:8.1f} us  write {wbytes/ms/1e6:7.1f} GB/s  alg {(wbytes+rbytes)/ms/1e6:7.1f} GB/s  {flops/ms/1e9:6.1f} TFLOP/s")

def bench_reduce(B, P):
    base = torch.randn(B, P, device="cuda"); clip = torch.rand(B, device="cuda")
    ws_b = nat.clip_reduce_workspace(B, P)
    ws = torch.empty(max(ws_b // 4, 4), device="cuda")
    summed = torch.zeros(P, device="cuda"); grad = torch.zeros(P, device="cuda")
    v = nat.FlatView(base, 0, P)
    ms = timeit(lambda: nat.clip_reduce_noise(v, clip, None, None, 1, 0, 1.0, 1.0 / B, summed, grad, B, ws))
    print(f"K7 B={B} P={P}: {ms*1e3:8.1f} us  read {B*P*4/ms/1e6:7.1f} GB/s (ws {ws_b>>20} MiB)")

def bench_gru(B, T, H, G=1):
    xps = [torch.randn(B, T, H, device="cuda") for _ in range(G)]
    xgs = [torch.randn(B, T, 3 * H, device="cuda") for _ in range(G)]
    W = [torch.randn(H, 2 * H, device="cuda") / (2 * H) ** 0.5 for _ in range(3 * G)]
    P = nat.gru_pack_forward(W, H)
    out, Z, R, N, Azr, An = nat.gru_forward(xps, xgs, None, *P)
    ms_f = timeit(lambda: nat.gru_forward(xps, xgs, None, *P))
    pb = nat.gru_pack_backward(W, H)
    douts = [torch.randn(B, T, H, device="cuda") for _ in range(G)]
    ms_b = timeit(lambda: nat.gru_backward(douts, None, Z, R, N, Azr, *pb, False))
    ms_p = timeit(lambda: (nat.gru_pack_forward(W, H), nat.gru_pack_backward(W, H)))
    flops = 2.0 * G * B * T * 3 * H * H   # hidden halves only; the input halves are ordinary GEMMs outside
    print(f"K8 G={G} B={B} T={T} H={H}: fwd {ms_f*1e3:8.1f} us ({flops/ms_f/1e9:5.1f} TFLOP/s)  bwd {ms_b*1e3:8.1f} us "
          f"({flops/ms_b/1e9:5.1f} TFLOP/s)  per step fwd {ms_f*1e3/T:6.2f} us bwd {ms_b*1e3/T:6.2f} us  packs {ms_p*1e3:5.1f} us")


def bench_lstm(B, T, H, G=2):
    xgs = [torch.randn(B, T, 4 * H, device="cuda") for _ in range(G)]
    W = [torch.randn(4 * H, H, device="cuda") / H ** 0.5 for _ in range(G)]
    b = [torch.zeros(4 * H, device="cuda") for _ in range(G)]
    P = nat.lstm_pack_forward(W, H); Pt = nat.lstm_pack_backward(W, H)
    rm = 2 if G > 1 else 0
    out, gates, C, Hprev = nat.lstm_forward(xgs, None, None, b, P, rm)
    ms_f = timeit(lambda: nat.lstm_forward(xgs, None, None, b, P, rm), iters=5, warm=1)
    dcat = torch.randn(B, T, G * H, device="cuda")
    douts = [dcat[..., g * H:(g + 1) * H] for g in range(G)]
    ms_b = timeit(lambda: nat.lstm_backward(douts, None, gates, C, Pt, rm, False), iters=5, warm=1)
    flops = 2.0 * G * B * T * 4 * H * H
    print(f"K10 G={G} B={B} T={T} H={H}: fwd {ms_f*1e3:8.1f} us ({flops/ms_f/1e9:5.1f} TFLOP/s)  bwd {ms_b*1e3:8.1f} us "
          f"({flops/ms_b/1e9:5.1f} TFLOP/s)  per step fwd {ms_f*1e3/T:6.2f} us bwd {ms_b*1e3/T:6.2f} us")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "k1"):
        for B in (64, 1024):
            bench_linear(B, 24, 400, 200)
            bench_linear(B, 24, 208, 208)
        bench_linear(1024, 24, 256, 128)
        bench_linear(1024, 24, 512, 512)
        bench_linear(1024, 24, 32, 128)
        bench_linear(512, 48, 250, 1000)
        bench_linear(512, 48, 1000, 250)
        bench_linear(512, 48, 250, 250)
        bench_linear(512, 48, 256, 1024)
        bench_linear(512, 48, 512, 1024)
        bench_linear(512, 48, 500, 1000)
        bench_linear(4096, 48, 250, 1000)
        bench_linear(4096, 48, 500, 1000)
        bench_linear(128, 256, 2048, 2048)
        bench_linear(256, 256, 1024, 512)
    if which in ("all", "k7"):
        bench_reduce(64, 649408)
        bench_reduce(1024, 824448)
        bench_reduce(2048, 2830272)
        bench_reduce(256, 12292096)
    if which in ("all", "k10"):
        bench_lstm(4096, 48, 250)
        bench_lstm(4096, 48, 256)
        bench_lstm(2048, 48, 250)
        bench_lstm(2048, 48, 256)
        bench_lstm(4096, 48, 128)
        bench_lstm(512, 48, 250)
    if which in ("all", "k8"):
        bench_gru(64, 24, 200)
        bench_gru(64, 24, 200, G=2)
        bench_gru(1024, 24, 128)
        bench_gru(1024, 24, 128, G=2)
        bench_gru(1024, 24, 200)
        bench_gru(4096, 24, 128)
        bench_gru(512, 256, 512)
