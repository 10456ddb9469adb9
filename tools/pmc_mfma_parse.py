"""Per-kernel matrix-pipe utilisation from tools/pmc_bench_mfma.sh output:  python tools/pmc_mfma_parse.py gpurun_out/mfma_swin_b_1dl
utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), summed over the last third of the launches (steady state)."""
import collections, csv, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(f"{d}/p_counter_collection.csv")))
per = collections.defaultdict(lambda: collections.defaultdict(float))
order = {}
for r in rows:
    key = (r["Kernel_Name"], r["Dispatch_Id"])
    per[key][r["Counter_Name"]] += float(r["Counter_Value"])
    order[key] = int(r["Dispatch_Id"])
keys = sorted(per, key=lambda k: order[k])
keys = keys[len(keys) * 2 // 3:]                       # steady state
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for k in keys:
    for c, v in per[k].items():
        agg[k[0]][c] += v
    agg[k[0]]["calls"] += 1
tot_cyc = sum(a["GRBM_GUI_ACTIVE"] for a in agg.values()) / 8
tot_mfma = sum(a["SQ_VALU_MFMA_BUSY_CYCLES"] for a in agg.values()) / 1024
print(f"steady-state sample: {len(keys)} dispatches, {tot_cyc:.3e} GPU cycles in kernels, matrix pipe busy {tot_mfma / tot_cyc * 100:.1f} % of them\n")
print("| kernel | calls | % of kernel cycles | matrix-pipe busy | f16 / bf16 MFMA Mops |\n|---|---|---|---|---|")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:16]:
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    print(f"| `{name[:70]}` | {int(a['calls'])} | {cyc / tot_cyc * 100:.1f} | {a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:.1f} % | "
          f"{a['SQ_INSTS_VALU_MFMA_MOPS_F16']:.3g} / {a['SQ_INSTS_VALU_MFMA_MOPS_BF16']:.3g} |")
