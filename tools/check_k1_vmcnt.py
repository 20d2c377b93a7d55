"""Static check of the hand-counted operand loads of linear_gs_split_kernel (csrc/k1_linear_gs.hip).

The kernel issues its operand loads from inline assembly (marker DPK_OPLOAD) and waits for them with an explicit
`s_waitcnt vmcnt(N)` (marker DPK_OPWAIT, which lists the registers it releases).  The compiler does not know that the
registers a DPK_OPLOAD writes are in flight, so this script checks in the generated ISA what the source promises:

  1. no instruction between a DPK_OPLOAD and the DPK_OPWAIT that releases its destination touches that destination;
  2. every path through a step issues the same number of vector-memory instructions (loads + stores) between two
     consecutive DPK_OPWAITs, so that N really is the number of younger operations (linear scan per inlined copy);
  3. the compiler itself inserted no `s_waitcnt vmcnt` (it would only cost time, but it means it saw a tracked load).

usage: python tools/check_k1_vmcnt.py [k1.s]   (without an argument: compiles csrc/k1_linear_gs.hip to ISA first)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, lines):
    inflight = {}
    errors = []
    n_load = n_wait = 0
    vm_since_wait = None
    per_copy = []
    for ln, raw in lines:
        code, _, comment = raw.partition(";")
        code = code.strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op, _, rest = code.partition(" ")
        if "DPK_OPLOAD" in comment:
            ops = [o.strip() for o in rest.split(",")]
            dst = regs(ops[0])
            for r in regs(",".join(ops[1:])) & set(inflight):
                errors.append(f"{name}:{ln}: operand load reads in-flight v{r}")
            for r in dst:
                inflight[r] = ln
            n_load += 1
            if vm_since_wait is not None:
                vm_since_wait += 1
            continue
        if "DPK_OPWAIT" in comment:
            released = regs(comment)
            missing = released - set(inflight)
            if missing and n_wait >= 2:
                errors.append(f"{name}:{ln}: wait releases registers with no load in flight: {sorted(missing)[:4]}...")
            for r in released:
                inflight.pop(r, None)
            n_wait += 1
            if vm_since_wait is not None:
                per_copy.append(vm_since_wait)
            vm_since_wait = 0
            continue
        if op == "s_waitcnt" and "vmcnt" in rest:
            errors.append(f"{name}:{ln}: compiler-inserted {code}")
        touched = regs(rest) & set(inflight)
        for r in touched:
            errors.append(f"{name}:{ln}: `{code}` touches v{r}, in flight since line {inflight[r]}")
        if vm_since_wait is not None and (op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_")):
            vm_since_wait += 1
    return n_load, n_wait, per_copy, errors


def check_private(name, lines, lo=120, hi=167):
    """linear_gs_ws_kernel: v[lo..hi] belong to the loaders' inline assembly (the kernel is compiled with
    amdgpu_num_vgpr(lo)); no instruction outside an ASMSTART/ASMEND block may name them.  Also lists the vector-memory
    instructions in the text between consecutive hand-counted waits."""
    errors, in_asm, vm, per = [], False, None, []
    for ln, raw in lines:
        if "ASMSTART" in raw:
            in_asm = True
            continue
        if "ASMEND" in raw:
            in_asm = False
            continue
        code, _, comment = raw.partition(";")
        code = code.strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        op, _, rest = code.partition(" ")
        if "DPK_PRIVWAIT" in comment:
            if vm is not None:
                per.append(vm)
            vm = 0
            continue
        if op == "s_waitcnt" and "vmcnt" in rest and not in_asm:
            errors.append(f"{name}:{ln}: compiler-inserted {code}")
        if op.startswith("scratch_"):
            errors.append(f"{name}:{ln}: spill `{code}` (a vector-memory instruction the hand-counted waits do not know)")
        if vm is not None and (op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_")):
            vm += 1
        if not in_asm:
            bad = [r for r in regs(rest) if lo <= r <= hi]
            if bad:
                errors.append(f"{name}:{ln}: `{code}` names private register v{bad[0]}")
    return per, errors


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "k1.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                        "--cuda-device-only", "-o", path, os.path.join(ROOT, "dp_timeautodiff_amd/csrc/k1_linear_gs.hip")],
                       check=True, stderr=subprocess.DEVNULL)
    kernels, cur, name = {}, None, None
    for ln, raw in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w*linear_gs_(?:split|wsl?)_kernel\w*):", raw)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append((ln, raw.rstrip("\n")))
            if "s_endpgm" in raw:
                cur = None
    bad = 0
    # the kernel descriptors must allocate the private registers (their clobber lists are what raises the count)
    text = open(path).read()
    for name in kernels:
        if "ws_kernel" in name or "wsl_kernel" in name:
            m = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\b.*?\.amdhsa_next_free_vgpr (\d+)", text, re.S)
            need = 256 if "wsl_kernel" in name else 168
            if not m or int(m.group(1)) < need:
                print(f"{name}: descriptor allocates {m.group(1) if m else '?'} VGPRs, the private registers need {need}")
                bad += 1
    for name, lines in kernels.items():
        if "ws_kernel" in name or "wsl_kernel" in name:
            per, errors = check_private(name, lines, lo=240 if "wsl_kernel" in name else 120, hi=255 if "wsl_kernel" in name else 167)
            ws = re.search(r"(wsl?)_kernelILi(\d+)E", name)
            print(f"{ws.group(1)}<{ws.group(2)}>: vector-memory instructions in the text between hand-counted waits: {per}, "
                  f"{len(errors)} problem(s)")
            for e in errors[:12]:
                print("   ", e)
            bad += len(errors)
            continue
        n_load, n_wait, per_copy, errors = check_kernel(name, lines)
        short = re.search(r"ILi(\d+)ELi(\d+)E", name)
        ws = re.search(r"ws_kernelILi(\d+)E", name)
        tag = f"split<{short.group(1)},{short.group(2)}>" if short else f"ws<{ws.group(1)}>" if ws else name
        print(f"{tag}: {n_load} operand loads, {n_wait} waits, vector-memory instructions in the text between waits: {per_copy}, "
              f"{len(errors)} problem(s)")
        for e in errors[:12]:
            print("   ", e)
        bad += len(errors)
    if not kernels:
        print("no linear_gs_split_kernel found")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
