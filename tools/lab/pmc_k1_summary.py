import collections, csv, glob, os, sys
out = sys.argv[1]
for lib in sys.argv[2:]:
    n = os.path.basename(lib)[:-3]
    vals = collections.defaultdict(list)
    for sub in ("sq", "grbm"):
        for f in glob.glob(os.path.join(out, f"pmc_{n}_{sub}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "linear_gs" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(os.path.join(out, f"kt_{n}", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "linear_gs" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = {k: sum(v[2:]) / max(len(v[2:]), 1) for k, v in vals.items()}
    d = sum(dur[2:]) / max(len(dur[2:]), 1)
    print(f"== {n}: {len(dur)} launches, mean {d:.1f} us (kernel trace)")
    for k in sorted(m):
        print(f"   {k:32s} {m[k]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
        print(f"   MFMA busy / SQ busy = {m['SQ_VALU_MFMA_BUSY_CYCLES']/m['SQ_BUSY_CYCLES']:.3f}")
    if "GRBM_GUI_ACTIVE" in m and d:
        print(f"   clock ~ GRBM_GUI_ACTIVE/8/time = {m['GRBM_GUI_ACTIVE']/8/d/1e3:.2f} GHz")
