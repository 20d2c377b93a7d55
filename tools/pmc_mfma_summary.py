"""Summarise the rocprofv3 passes of tools/pmc_mfma.sh into profiles/<tag>_pmc_mfma.{md,json}: per hand-written kernel
(hot-path replay of one config) the launch count, the mean duration from the kernel trace, the SQ MFMA counters, the
clock GRBM_GUI_ACTIVE implies, and HBM traffic (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md §HBM).

Derived columns:
  clock        = GRBM_GUI_ACTIVE / 8 XCDs / duration          (the shader clock the launch really ran at)
  pipe         = which matrix pipe the kernel's MFMAs use: "f32" when SQ_INSTS_VALU_MFMA_MOPS_F32 > 0, else "bf16"
                 (linear_gs_split_kernel<TERMS, MODE>: TERMS bf16 partial products per fp32 product)
  pipe TFLOP/s = SQ_VALU_MFMA_BUSY_CYCLES x (64 | 1024) / duration   (f32: 32x32x2 = 4096 flop in 64 busy cycles, 16x16x4 =
                 2048 in 32: 64 per busy cycle; bf16: 32x32x16 = 32768 flop in 32 busy cycles: 1024)
  fp32 TFLOP/s = pipe TFLOP/s for f32 kernels; pipe TFLOP/s / TERMS for the split kernel (algorithmic fp32 flops)
  vs peak      = pipe TFLOP/s / (157.3 f32 | 2516.6 bf16 dense peak at 2.4 GHz)
  pipe busy    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)    share of SIMD-cycles with an MFMA in flight
usage: python3 tools/pmc_mfma_summary.py <dir> <tag> "<command>"
       python3 tools/pmc_mfma_summary.py --rederive profiles/<tag>_pmc_mfma.json      (recompute the derived columns)"""
import collections, csv, glob, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mean = lambda v: sum(v) / len(v) if v else 0.0


def short(k):
    if "(anonymous namespace)::" not in k or "at::native" in k:
        return None
    return k.split("(anonymous namespace)::")[1].split("(")[0]


def counters(d, sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def collect(d):
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    sq, grbm, fetch, write = (counters(d, s) for s in ("sq", "grbm", "fetch", "write"))
    rows = {}
    for k in dur:
        c = {n: mean(v) for n, v in sq.get(k, {}).items()}
        c.update({n: mean(v) for n, v in grbm.get(k, {}).items()})
        rows[k] = {"launches": len(dur[k]), "mean_us": mean(dur[k]), "total_us": sum(dur[k]), "counters": c,
                   "fetch_bytes": mean(fetch.get(k, {}).get("FETCH_SIZE", [])) * 2 * 1024,
                   "write_bytes": mean(write.get(k, {}).get("WRITE_SIZE", [])) * 1024}
    return rows


def derive(rows):
    for k, r in rows.items():
        c, us = r["counters"], r["mean_us"]
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        bf16 = busy > 0 and c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) == 0
        m = re.match(r"linear_gs_(?:split|wsl?)_kernel<(\d+)", k)
        terms = int(m.group(1)) if (m and bf16) else 1
        r["pipe"] = "bf16" if bf16 else ("f32" if busy > 0 else "-")
        r["clock_GHz"] = gui / 8 / us / 1e3 if us and gui else None
        r["pipe_tflops"] = busy * (1024 if bf16 else 64) / us / 1e6 if us else 0.0
        r["fp32_tflops"] = r["pipe_tflops"] / terms
        r["pipe_vs_peak"] = r["pipe_tflops"] / (2516.6 if bf16 else 157.3)
        r["mfma_pipe_busy"] = busy / (1024 * gui / 8) if gui else None
        r["hbm_TBs"] = (r["fetch_bytes"] + r["write_bytes"]) / us / 1e6 if us else 0.0
    return rows


def write(tag, what, rows):
    json.dump({"command": what, "note": __doc__, "kernels": rows}, open(os.path.join(root, "profiles", f"{tag}_pmc_mfma.json"), "w"), indent=1)
    path = os.path.join(root, "profiles", f"{tag}_pmc_mfma.md")
    with open(path, "w") as f:
        f.write(f"# {tag}: rocprofv3 --pmc passes (SQ, GRBM, FETCH_SIZE, WRITE_SIZE) + --kernel-trace over `{what}`\n\n")
        f.write("per-launch means; pipe TFLOP/s = SQ_VALU_MFMA_BUSY_CYCLES x 64 (f32 pipe) or x 1024 (bf16 pipe) / duration; fp32 "
                "TFLOP/s = algorithmic fp32 flops (the split kernel issues TERMS bf16 MFMAs per fp32 one); clock = "
                "GRBM_GUI_ACTIVE/8/duration; pipe busy = MFMA busy cycles / (1024 SIMDs x active cycles); peaks 157.3 (f32) / "
                "2516.6 (bf16) TFLOP/s, 8 TB/s HBM; fetch = FETCH_SIZE x 2 (gfx950 correction)\n\n")
        f.write("| kernel | launches | mean us | clock GHz | pipe | MFMA busy cyc | MOPS_F32 | SQ busy cyc | pipe TFLOP/s | vs pipe peak | fp32 TFLOP/s | pipe busy | fetch MB | write MB | HBM TB/s |\n")
        f.write("|" + "---|" * 15 + "\n")
        for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_us"]):
            c = r["counters"]
            f.write(f"| `{k}` | {r['launches']} | {r['mean_us']:.1f} | {r['clock_GHz'] or 0:.2f} | {r['pipe']} | "
                    f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g} | {c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):.3g} | "
                    f"{c.get('SQ_BUSY_CYCLES', 0):.3g} | {r['pipe_tflops']:.1f} | {r['pipe_vs_peak']:.3f} | {r['fp32_tflops']:.1f} | "
                    f"{r['mfma_pipe_busy'] or 0:.3f} | {r['fetch_bytes']/1e6:.1f} | {r['write_bytes']/1e6:.1f} | {r['hbm_TBs']:.2f} |\n")
    print(open(path).read())


if sys.argv[1] == "--rederive":
    j = json.load(open(sys.argv[2]))
    tag = os.path.basename(sys.argv[2])[:-len("_pmc_mfma.json")]
    write(tag, j["command"], derive(j["kernels"]))
else:
    write(sys.argv[2], sys.argv[3], derive(collect(sys.argv[1])))
