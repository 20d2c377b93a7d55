"""lab: where does the role-split kernel differ from the split kernel?  usage: python tools/lab/ws_diff.py B T in out"""
import sys
import torch
sys.path.insert(0, ".")
from dp_timeautodiff_amd import _native as nat
nat.load()
B, T, in_f, out_f = (int(v) for v in sys.argv[1:5])
g = torch.Generator().manual_seed(1)
A = torch.randn(B, T, in_f, generator=g).cuda()
G = (torch.randn(B, T, out_f, generator=g) * 0.05).cuda()
ld = (in_f + 31) // 32 * 32 if in_f % 32 else in_f
win = (ld * out_f + 63) // 64 * 64
bw = (out_f + 63) // 64 * 64
slots = nat.linear_slots(in_f, out_f)
outs = {}
for split in (False, True):
    nat.linear_configure(arithmetic="split6", role_split=split)
    base = torch.full((B, win + bw + 64), 7.0, device="cuda")
    part = torch.full((B, slots + 3), -1.0, device="cuda")
    nat.linear_grad_sample(A, G, nat.FlatView(base, 0, ld * out_f), nat.FlatView(base, win, out_f), part, 0, float(B), ld_w=ld)
    torch.cuda.synchronize()
    outs[split] = (base, part)
w0 = outs[False][0][:, :ld * out_f].reshape(B, out_f, ld)
w1 = outs[True][0][:, :ld * out_f].reshape(B, out_f, ld)
d = (w0 != w1)
print("weight mismatches:", int(d.sum()), "of", d.numel())
if d.any():
    idx = d.nonzero()
    print("samples:", sorted(set(idx[:, 0].tolist()))[:20])
    print("rows (mod 128 tile):", sorted(set((idx[:, 1] // 128).tolist())), "first rows", sorted(set(idx[:, 1].tolist()))[:20])
    print("cols tile:", sorted(set((idx[:, 2] // 128).tolist())), "first cols", sorted(set(idx[:, 2].tolist()))[:20])
    b, r, c = idx[0].tolist()
    print("example", b, r, c, float(w0[b, r, c]), float(w1[b, r, c]))
b0 = outs[False][0][:, win:win + out_f]; b1 = outs[True][0][:, win:win + out_f]
print("bias mismatches:", int((b0 != b1).sum()))
print("part mismatches:", int((outs[False][1] != outs[True][1]).sum()), "tail ok:", float(outs[True][0][:, -64:].min()))
if (b0 != b1).any():
    ref = (G.double() * B).sum(1)
    idx = (b0 != b1).nonzero()
    print("samples with bias mismatches:", sorted(set(idx[:, 0].tolist())))
    print("features (first 24):", sorted(set(idx[:, 1].tolist()))[:24])
    for k in range(0, min(len(idx), 6)):
        b, f = idx[k].tolist()
        print(b, f, "split", float(b0[b, f]), "ws", float(b1[b, f]), "ref", float(ref[b, f]))
    print("max rel err split", float(((b0.double() - ref).abs() / ref.abs().amax()).max()), "ws", float(((b1.double() - ref).abs() / ref.abs().amax()).max()))
pm = (outs[False][1] != outs[True][1]).nonzero()
tiles_n = (in_f + 127) // 128
for k in range(min(len(pm), 12)):
    b, sl = pm[k].tolist()
    print("part mismatch sample", b, "slot", sl, "= (tm, tn)", divmod(sl, tiles_n), "split", float(outs[False][1][b, sl]), "ws", float(outs[True][1][b, sl]))
