"""Where do two K1 builds differ? (lab)  usage: python3 tools/lab/k1_diff.py a.so b.so T in out B"""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_void_p
import torch
T, in_f, out_f, B = (int(v) for v in sys.argv[3:7])
ld = (in_f + 31) // 32 * 32 if (in_f >= 64 and in_f % 32) else in_f
win = (ld * out_f + 63) // 64 * 64
row = win + (out_f + 63) // 64 * 64
torch.manual_seed(0)
A = torch.randn(B, T, in_f, device="cuda"); G = torch.randn(B, T, out_f, device="cuda")
st = c_void_p(torch.cuda.current_stream().cuda_stream)
res = []
for path in sys.argv[1:3]:
    lib = ctypes.CDLL(os.path.abspath(path))
    f = lib.dpk_linear_grad_sample
    f.restype = c_int
    f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int, c_float, c_void_p]
    lib.dpk_linear_slots.restype = c_int
    arena = torch.full((B, row), 7.0, device="cuda")
    part = torch.zeros(B, lib.dpk_linear_slots(in_f, out_f) + 4, device="cuda")
    rc = f(A.data_ptr(), G.data_ptr(), arena.data_ptr(), arena.data_ptr() + 4 * win, part.data_ptr(), part.stride(0), 0, B, T, in_f, out_f, row, ld, float(B), st)
    assert rc == 0
    torch.cuda.synchronize()
    res.append((arena, part))
a, b = res[0][0], res[1][0]
d = (a - b).abs()
rel = d / a.abs().clamp_min(1.0)
print("max abs", float(d.max()), "max rel", float(rel.max()), "typical |a|", float(a.abs().mean()))
idx = torch.nonzero(rel > 1e-4)
print("n bad", idx.shape[0])
if idx.shape[0]:
    for r, c in idx[:10].tolist():
        where = "bias" if c >= win else f"w[{c // ld},{c % ld}]"
        print(r, c, where, float(a[r, c]), float(b[r, c]))
    print("bad rows", torch.unique(idx[:, 0])[:20].tolist(), "bad col range", int(idx[:, 1].min()), int(idx[:, 1].max()))
print("part rel", float(((res[0][1] - res[1][1]).abs() / res[0][1].abs().clamp_min(1)).max()))
