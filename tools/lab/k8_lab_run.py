"""K8 lab: cycle breakdown of one workgroup's forward time step (wave 0), from clock64() stamps."""
import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import torch
from dp_timeautodiff_amd import _native as nat
nat.LIB_PATH = os.path.join(here, "libdpk_lab.so")
lib = nat.load()
names = ["x->LDS", "barrier0", "phase1 mfma", "phase1 epilogue", "barrier1", "phase2 mfma", "phase2 epilogue", "barrier2"]
for (B, T, H) in ((1024, 24, 128), (1024, 24, 200)):
    xp = [torch.randn(B, T, H, device="cuda")]
    W = [torch.randn(H, 2 * H, device="cuda") / (2 * H) ** 0.5 for _ in range(3)]
    b = [torch.zeros(H, device="cuda") for _ in range(3)]
    P = nat.gru_pack_forward(W, H)
    for _ in range(3):
        nat.gru_forward(xp, None, *P, b)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 128)()
    lib.k8_dbg_read(buf)
    tot = sum(buf[:8])
    print(f"B={B} T={T} H={H}: total {tot} ticks over {T} steps = {tot/T:.0f}/step")
    for n, v in zip(names, buf[:8]):
        print(f"   {n:18s} {v/T:9.0f} ticks/step  {100*v/tot:5.1f}%")
    t0 = min(buf[16 + w * 8] for w in range(8))
    print("   per-wave stamps at t=5 (ticks after the earliest wave reached barrier0):")
    for w in range(8):
        print("   wave", w, " ".join(f"{buf[16 + w * 8 + n] - t0:7d}" for n in range(8)))
