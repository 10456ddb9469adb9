import collections, csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "split_linear"
for d in sorted(glob.glob("gpurun_out/gm_*/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:45s} {sum(v)/len(v):16.0f}")
