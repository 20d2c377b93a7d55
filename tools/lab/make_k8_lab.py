"""Generate tools/lab/k8_lab.hip: the product K8 source with clock64() stamps around the phases of the forward
time step (wave 0 of workgroup 0), read back through k8_dbg_read().  Lab only; never shipped."""
import os
here = os.path.dirname(os.path.abspath(__file__))
s = open(os.path.join(here, "../../dp_timeautodiff_amd/csrc/k8_gru.hip")).read()
s = s.replace('"dpk_common.h"', '"../../dp_timeautodiff_amd/csrc/dpk_common.h"')
s = s.replace('namespace {', '__device__ long long k8_dbg[128];\nextern "C" int k8_dbg_read(long long* h) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(k8_dbg), sizeof(long long) * 128); }\n#define TICK(n) do { long long now_ = clock64(); if (dbg) acc_[n] += now_ - last_; last_ = now_; if (blockIdx.x == 0 && lane == 0 && t == 5) k8_dbg[16 + wave * 8 + n] = now_; } while (0)\nnamespace {', 1)
a = s.index('template <int TPW, bool PRE>\n__global__ void __launch_bounds__(GNW * 64) gru_fwd_kernel')
b = s.index('template <int TPW, bool PRE>\n__global__ void __launch_bounds__(GNW * 64) gru_bwd_kernel')
f = s[a:b]


def rep(old, new):
    global f
    assert old in f, old
    f = f.replace(old, new)


rep('    if (PRE) load_x(0);\n', '    if (PRE) load_x(0);\n    const bool dbg = blockIdx.x == 0 && threadIdx.x == 0;\n    long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};\n    long long last_ = clock64();\n')
rep('        lds_barrier();\n        if (PRE && t + 1 < T) load_x(t + 1);', '        TICK(0);\n        lds_barrier();\n        TICK(1);\n        if (PRE && t + 1 < T) load_x(t + 1);')
rep('bufA, KG, lane);\n', 'bufA, KG, lane);\n                asm volatile("s_nop 0" :: "v"(az.x), "v"(ar.x));\n                TICK(2);\n')
rep('        lds_barrier();\n        // ---- phase 2', '        TICK(3);\n        lds_barrier();\n        TICK(4);\n        // ---- phase 2')
rep('bufB, KG, lane);\n', 'bufB, KG, lane);\n                asm volatile("s_nop 0" :: "v"(an.x));\n                TICK(5);\n')
rep("        lds_barrier();  // bufB's x rows are rewritten at the top of the next step\n    }\n", "        TICK(6);\n        lds_barrier();  // bufB's x rows are rewritten at the top of the next step\n        TICK(7);\n    }\n    if (dbg) for (int i = 0; i < 8; ++i) k8_dbg[i] = acc_[i];\n")
s = s[:a] + f + s[b:]
open(os.path.join(here, "k8_lab.hip"), "w").write(s)
