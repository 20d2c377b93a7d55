"""Condense a rocprofv3 --kernel-trace --stats output dir into profiles/<tag>_kernel_stats.csv (+ .md)."""
import csv, glob, os, sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
out_csv = os.path.join(root, "profiles", f"{tag}_kernel_stats.csv")
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        name = r["Name"]
        short = name if len(name) < 160 else name[:157] + "..."
        w.writerow([short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
ours = [r for r in rows if "anonymous namespace" in r["Name"] and any(k in r["Name"] for k in
        ("linear_gs", "clip_reduce", "finalize", "clip_factor", "embedding_gs", "layernorm_gs", "noise_scale", "normsq", "adam", "gru", "lstm", "poisson"))]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({tag})\n\n")
    f.write(f"command: {' '.join(sys.argv[3:]) if len(sys.argv) > 3 else 'python3 bench.py'}\n\n")
    f.write(f"total GPU kernel time {tot/1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} dispatches\n\n")
    f.write("## dpk kernels\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in ours:
        f.write(f"| `{r['Name'].split('(anonymous namespace)::')[1].split('(')[0]}` | {r['Calls']} | "
                f"{float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']} |\n")
    f.write("\n## top 12 overall\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
    for r in rows[:12]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']} |\n")
print(open(os.path.join(root, "profiles", f"{tag}_kernel_stats.md")).read())
