"""From a rocprofv3 kernel trace CSV: busy time vs wall span of the last N graph-replayed steps, top kernels per step.
A step is delimited by consecutive launches of clip_factor_kernel (one per DP-SGD step)."""
import csv, glob, sys, collections
src = sys.argv[1]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "clip_factor_kernel" in r["Kernel_Name"]]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    span = int(rows[b]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    steps.append((span, busy, len(seg), seg))
steps = [s for s in steps if s[2] > 50]
tail = steps[-10:]
print(f"{len(steps)} steps; last {len(tail)}: span {sum(s[0] for s in tail)/len(tail)/1e6:.3f} ms, busy {sum(s[1] for s in tail)/len(tail)/1e6:.3f} ms, {tail[-1][2]} kernels/step")
agg = collections.defaultdict(lambda: [0, 0])
for s in tail:
    for r in s[3]:
        n = r["Kernel_Name"]
        n = n.split("(anonymous namespace)::")[-1][:70] if "anonymous" in n else n[:70]
        agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[n][1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{t/len(tail)/1e3:9.1f} us/step {c/len(tail):6.1f} calls  {n}")
