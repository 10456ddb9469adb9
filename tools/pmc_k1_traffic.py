"""gpurun_out/k1_{FETCH_SIZE,WRITE_SIZE}/p_*.csv -> profiles/k1_pmc.json (+ copies of the counter CSVs)."""
import csv, glob, json, shutil, sys
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
TAG = f"_{H}x{W}" if len(sys.argv) > 2 else ""
Q, K, HW = 100, 19, H * W


def _one(pattern):
    return sorted(glob.glob(pattern, recursive=True))[0]


vals, durs = {}, []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open(_one(f"gpurun_out/k1{TAG}_{c}/**/*counter_collection.csv"))) if "rba_reduce_pk_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    per = {}
    for r in rows:
        per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals[c] = sum(per.values()) / len(per)
    tr = [r for r in csv.DictReader(open(_one(f"gpurun_out/k1{TAG}_{c}/**/*kernel_trace.csv"))) if "rba_reduce_pk_kernel" in r["Kernel_Name"]]
    durs += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
fetch, write = vals["FETCH_SIZE"] * 1024 * 2.0, vals["WRITE_SIZE"] * 1024
alg = 4 * Q * HW + 4 * Q * K + 4 * HW
out = {
    "kernel": "rba_reduce_pk_kernel<19,false,false,2,4,true> (rba_reduce_ws_f32, the product path)",
    "workload": f"mask_pred N(0,25) [100,{H},{W}] fp32, cls_prob [100,19] (tools/k1_sweep.py 121{' --hw %d %d' % (H, W) if TAG else ''})", "round": int(__import__("os").environ.get("RBA_ROUND", "6")),
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python tools/k1_sweep.py 121 (one counter per pass; tools/pmc_k1_traffic.sh)",
    "FETCH_SIZE_raw_KB": vals["FETCH_SIZE"], "WRITE_SIZE_raw_KB": vals["WRITE_SIZE"], "fetch_correction": 2.0,
    "fetch_correction_reason": "gfx950 rocprofv3: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while wide coalesced reads are 128-B requests (MI355X_MICROARCH.md section HBM); WRITE_SIZE calibrates exactly (4*H*W bytes)",
    "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": alg,
    "avg_launch_us_during_pmc": sum(durs) / len(durs), "launches": len(durs), "traffic_over_algorithmic": (fetch + write) / alg,
}
json.dump(out, open(f"profiles/k1_pmc{TAG}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
