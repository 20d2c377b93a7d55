"""Run a DPK_K1_STAMPS lab build once and print the per-phase cycle shares (mean over waves of the first 64 WGs)."""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_void_p
import numpy as np, torch
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
T, in_f, out_f = (int(v) for v in sys.argv[2:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
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
for i in range(4):
    rc = f(A.data_ptr(), G.data_ptr(), arena.data_ptr(), arena.data_ptr() + 4 * win, part.data_ptr(), part.stride(0), 0, B, T, in_f, out_f, row, ld, float(B), st)
    assert rc == 0, rc
torch.cuda.synchronize()
NW = int(os.environ.get('NWAVES_WG', '8'))
n = 64 * NW * 9
host = (ctypes.c_ulonglong * n)()
lib.dpk_lab_read_stamps.argtypes = [c_void_p, c_int]
assert lib.dpk_lab_read_stamps(host, n) == 0
a = np.array(host[:], dtype=np.float64).reshape(64, NW, 9)
NWV = int(os.environ.get('NWAVES', str(NW)))
a = a[:, :NWV]
names = sys.argv[6].split(",") if len(sys.argv) > 6 else [f"p{i}" for i in range(8)]
items = a[..., 8]
tot = a[..., :7].sum(-1)
print(f"items per WG: {items.mean():.1f}; cycles per item per wave: {(tot / items).mean():.0f}")
for w in range(NWV):
    print(f"wave {w}: " + "  ".join(f"{names[i]}={a[:, w, i].sum() / items[:, w].sum():7.0f}" for i in range(8)))
print(f"100 MHz ticks per item: {(a[..., 7] / items).mean():.1f}  -> core clock {(tot / a[..., 7]).mean() * 0.1:.2f} GHz if s_memtime counts core cycles")
print("all   : " + "  ".join(f"{names[i]}={a[..., i].sum() / items.sum():7.0f}" for i in range(8)))
