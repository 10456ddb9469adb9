"""Summarise tools/pmc_k2.sh: per-launch averages of every counter for the K2 kernels.  python tools/pmc_k2_parse.py [form]"""
import collections, csv, glob, sys
form = sys.argv[1] if len(sys.argv) > 1 else "fused"
dur = []
for d in sorted(glob.glob(f"gpurun_out/k2pmc_{form}_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "msda_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:40s} {sum(v) / len(v):18.0f}   ({len(v)} launches)")
for d in sorted(glob.glob(f"gpurun_out/k2pmc_{form}_*/**/*kernel_trace.csv", recursive=True))[:1]:
    for r in csv.DictReader(open(d)):
        if "msda_fwd" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if dur:
    print(f"kernel duration under the profiler: median {sorted(dur)[len(dur) // 2]:.1f} us over {len(dur)} launches")
