"""K8 lab: time the recurrence kernels with parts of their memory traffic switched off (flags: 1 stores, 2 operand
prefetch loads, 4 weight loads).  Results of the reduced variants are wrong on purpose; only the timing is read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dp_timeautodiff_amd import _native as nat
nat.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libdpk_exp{sys.argv[1]}.so")
sys.argv = [sys.argv[0], "none"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kbench
print("flags", nat.LIB_PATH[-8:])
kbench.bench_gru(1024, 24, 128)
kbench.bench_gru(1024, 24, 200)
