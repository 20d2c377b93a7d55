"""A/B timing of K1 builds (lab).  Each build is a shared object that exports dpk_linear_grad_sample_group /
dpk_linear_slots (the product library, or `hipcc -shared` of csrc/k1_linear_gs.hip alone with -D knobs).

usage: python tools/lab/k1_ab.py [--shapes cfg3|cfg2|cfg1|all] [--B N] lib_a.so lib_b.so ...
Builds are timed interleaved (a b c a b c ...) in rounds of `--iters` launches, at real-step sizes (operands and
arena beyond the 256 MB Infinity Cache), and every build's output is compared with the first one's.
"""
import argparse
import ctypes
import os
import sys
from ctypes import c_float, c_int, c_int64, c_void_p

import torch

SHAPES = {
    # (T, in, out, n_jobs)
    "cfg3": [(48, 250, 1000, 1), (48, 500, 1000, 1), (48, 1000, 250, 1), (48, 250, 250, 1)],
    "cfg2": [(24, 512, 512, 1), (24, 256, 128, 6), (24, 512, 128, 1)],
    "cfg1": [(24, 400, 200, 6), (24, 208, 208, 1)],
    "lab": [(48, 512, 512, 1), (24, 512, 512, 1)],
    "cfg5": [(256, 512, 512, 1), (256, 1024, 512, 1)],
    "cfg5big": [(256, 2048, 2048, 1), (256, 1024, 512, 6)],   # run with --B 2048 (one physical chunk): 34 / 25 GB arenas
}


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.dpk_linear_slots.restype = c_int
    lib.dpk_linear_slots.argtypes = [c_int, c_int]
    f = lib.dpk_linear_grad_sample_group
    f.restype = c_int
    f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int,
                  c_int, c_int64, c_int, c_int, c_float, c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--shapes", default="cfg3")
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--bias", type=int, default=1)
    args = ap.parse_args()
    # "path.so@mode" runs that library with DPK_K1_MFMA=mode (f32 | split6 | split8 | split9)
    modes = {}
    libs = []
    for p in args.libs:
        path, _, mode = p.partition("@")
        name = os.path.basename(path) + ("@" + mode if mode else "")
        modes[name] = mode
        libs.append((name, load(path)))
    names = sum((SHAPES[k] for k in (SHAPES if args.shapes == "all" else args.shapes.split(","))), [])
    stream = c_void_p(torch.cuda.current_stream().cuda_stream)
    for T, in_f, out_f, nj in names:
        B = args.B
        ld = (in_f + 31) // 32 * 32 if (in_f >= 64 and in_f % 32) else in_f
        win = (ld * out_f + 63) // 64 * 64
        row = nj * (win + (out_f + 63) // 64 * 64)
        torch.manual_seed(0)
        A = [torch.randn(B, T, in_f, device="cuda") for _ in range(nj)]
        G = [torch.randn(B, T, out_f, device="cuda") for _ in range(nj)]
        slots = libs[0][1].dpk_linear_slots(in_f, out_f)
        outs = []
        res = {}
        arena = torch.zeros(B, row, device="cuda")
        part = torch.zeros(B, nj * slots + 4, device="cuda")
        arr = lambda vals: (c_void_p * nj)(*vals)
        pa, pg = arr([a.data_ptr() for a in A]), arr([g.data_ptr() for g in G])
        pw = arr([arena.data_ptr() + 4 * j * (win + (out_f + 63) // 64 * 64) for j in range(nj)])
        pb = arr([arena.data_ptr() + 4 * (j * (win + (out_f + 63) // 64 * 64) + win) for j in range(nj)])
        sl = (c_int * nj)(*[j * slots for j in range(nj)])

        def run(lib, name=None):
            if name is not None and hasattr(lib, "dpk_linear_configure"):
                # "lib.so@split8,ws0": arithmetic and / or role split through the library's own switch
                arith, ws = -1, -1
                for tok in filter(None, modes.get(name, "").split(",")):
                    if tok in ("ws0", "ws1"):
                        ws = int(tok[2])
                    else:
                        arith = {"f32": 0, "split6": 6, "split8": 8, "split9": 9}[tok]
                lib.dpk_linear_configure(arith, ws)
            elif name is not None and modes.get(name):
                os.environ["DPK_K1_MFMA"] = modes[name]
            rc = lib.dpk_linear_grad_sample_group(pa, pg, pw, pb if args.bias else None, sl, nj, c_void_p(part.data_ptr()),
                                                  part.stride(0), B, T, in_f, out_f, row, ld, out_f, float(B), stream)
            if rc:
                raise SystemExit(f"rc={rc}")

        times = {n: [] for n, _ in libs}
        for n, lib in libs:  # correctness first
            arena.zero_(); part.zero_()
            run(lib, n)
            torch.cuda.synchronize()
            outs.append((n, arena[:64].clone(), arena[-8:].clone(), part.clone()))
        for _ in range(args.rounds):
            for n, lib in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run(lib, n)
                e0.record()
                for _ in range(args.iters):
                    run(lib)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / args.iters)
        wbytes = B * nj * (ld * out_f + out_f) * 4.0
        rbytes = B * nj * T * (in_f + out_f) * 4.0
        flops = 2.0 * B * nj * T * in_f * out_f
        print(f"B={B} T={T} in={in_f} out={out_f} jobs={nj}  ({wbytes/1e9:.2f} GB written)")
        for n, _ in libs:
            ts = sorted(times[n])
            ms = ts[len(ts) // 2]
            print(f"  {n:28s} {ms*1e3:8.1f} us (min {ts[0]*1e3:8.1f})  {(wbytes+rbytes)/ms/1e9:7.2f} TB/s alg  "
                  f"{flops/ms/1e9:6.1f} TFLOP/s")
        ref = outs[0]
        for n, a0, a1, p in outs[1:]:
            d = max(float((a0 - ref[1]).abs().max()), float((a1 - ref[2]).abs().max()))
            dp = float((p.double().sum(1) - ref[3].double().sum(1)).abs().max() / ref[3].double().sum(1).abs().max())
            print(f"  {n:28s} max |diff| vs {ref[0]} = {d:.3g}; normsq rel diff {dp:.2e}")
        del A, G, arena, part
        torch.cuda.empty_cache()
        sys.stdout.flush()


if __name__ == "__main__":
    main()
