"""One K1 build, one shape, a few launches — the program rocprofv3 --pmc wraps (lab).
usage: python3 tools/lab/k1_one.py lib.so T in out [B] [iters]"""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_void_p
import torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
T, in_f, out_f = (int(v) for v in sys.argv[2:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
f = lib.dpk_linear_grad_sample
f.restype = c_int
f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int, c_float, c_void_p]
lib.dpk_linear_slots.restype = c_int
ld = (in_f + 31) // 32 * 32 if (in_f >= 64 and in_f % 32) else in_f
win = (ld * out_f + 63) // 64 * 64
row = win + (out_f + 63) // 64 * 64
A = torch.randn(B, T, in_f, device="cuda"); G = torch.randn(B, T, out_f, device="cuda")
arena = torch.zeros(B, row, device="cuda")
part = torch.zeros(B, lib.dpk_linear_slots(in_f, out_f) + 4, device="cuda")
st = c_void_p(torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(iters + 2):
    if i == 2: e0.record()
    rc = f(A.data_ptr(), G.data_ptr(), arena.data_ptr(), arena.data_ptr() + 4 * win, part.data_ptr(), part.stride(0), 0, B, T, in_f, out_f, row, ld, float(B), st)
    assert rc == 0, rc
e1.record(); torch.cuda.synchronize()
print(f"{os.path.basename(sys.argv[1])} T={T} {in_f}->{out_f} B={B}: {e0.elapsed_time(e1)/iters*1e3:.1f} us")
