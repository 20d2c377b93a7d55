"""Summarise the rocprofv3 passes of tools/pmc_mfma.sh into profiles/<tag>_pmc_mfma.{md,json}: per hand-written kernel
(hot-path replay of one config) the launch count, the mean duration from the kernel trace, the SQ MFMA counters, the
clock GRBM_GUI_ACTIVE implies, and HBM traffic (FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md §HBM).

Derived columns:
  clock       = GRBM_GUI_ACTIVE / 8 XCDs / duration           (the shader clock the launch really ran at)
  MFMA flops  = SQ_VALU_MFMA_BUSY_CYCLES x 64                  (fp32 MFMA: 32x32x2 = 4096 flop in 64 busy cycles,
                                                               16x16x4 = 2048 flop in 32: 64 flop per busy cycle)
  MFMA pipe   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   share of SIMD-cycles with an MFMA in flight
  vs peak     = MFMA flops / duration / 157.3 TFLOP/s          (fp32 dense MFMA peak at 2.4 GHz, the roofline's number)
usage: python3 tools/pmc_mfma_summary.py <dir> <tag> "<command>" """
import collections, csv, glob, json, os, sys

d, tag, what = sys.argv[1], sys.argv[2], sys.argv[3]


def short(k):
    if "(anonymous namespace)::" not in k or "at::native" in k:
        return None
    return k.split("(anonymous namespace)::")[1].split("(")[0]


def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
mean = lambda v: sum(v) / len(v) if v else 0.0
rows = {}
sq, grbm, fetch, write = counters("sq"), counters("grbm"), counters("fetch"), counters("write")
for k in dur:
    c = {n: mean(v) for n, v in sq.get(k, {}).items()}
    c.update({n: mean(v) for n, v in grbm.get(k, {}).items()})
    us = mean(dur[k])
    busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
    row = {"launches": len(dur[k]), "mean_us": us, "total_us": sum(dur[k]), "counters": c,
           "clock_GHz": gui / 8 / us / 1e3 if us and gui else None,
           "mfma_tflops": busy * 64 / us / 1e6 if us else 0.0,
           "mfma_pipe_busy": busy / (1024 * gui / 8) if gui else None,
           "fetch_bytes": mean(fetch.get(k, {}).get("FETCH_SIZE", [])) * 2 * 1024,
           "write_bytes": mean(write.get(k, {}).get("WRITE_SIZE", [])) * 1024}
    row["mfma_vs_peak"] = row["mfma_tflops"] / 157.3
    row["hbm_TBs"] = (row["fetch_bytes"] + row["write_bytes"]) / us / 1e6 if us else 0.0
    rows[k] = row
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump({"command": what, "note": __doc__, "kernels": rows}, open(os.path.join(root, "profiles", f"{tag}_pmc_mfma.json"), "w"), indent=1)
with open(os.path.join(root, "profiles", f"{tag}_pmc_mfma.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 --pmc passes (SQ, GRBM, FETCH_SIZE, WRITE_SIZE) + --kernel-trace over `{what}`\n\n")
    f.write("per-launch means; MFMA flops = SQ_VALU_MFMA_BUSY_CYCLES x 64; clock = GRBM_GUI_ACTIVE/8/duration; "
            "pipe busy = MFMA busy cycles / (1024 SIMDs x active cycles); peak 157.3 TFLOP/s fp32 MFMA, 8 TB/s HBM\n\n")
    f.write("| kernel | launches | mean us | clock GHz | MFMA busy cyc | MFMA MOPS_F32 | SQ busy cyc | MFMA TFLOP/s | vs peak | pipe busy | fetch MB | write MB | HBM TB/s |\n")
    f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_us"]):
        c = r["counters"]
        f.write(f"| `{k}` | {r['launches']} | {r['mean_us']:.1f} | {r['clock_GHz'] or 0:.2f} | "
                f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g} | {c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):.3g} | "
                f"{c.get('SQ_BUSY_CYCLES', 0):.3g} | {r['mfma_tflops']:.1f} | {r['mfma_vs_peak']:.3f} | "
                f"{r['mfma_pipe_busy'] or 0:.3f} | {r['fetch_bytes']/1e6:.1f} | {r['write_bytes']/1e6:.1f} | {r['hbm_TBs']:.2f} |\n")
print(open(os.path.join(root, "profiles", f"{tag}_pmc_mfma.md")).read())
