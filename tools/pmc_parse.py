"""Summarise gpurun_out/gp_{a,b,c}/p_counter_collection.csv for one kernel-name substring."""
import collections, csv, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "split_linear"
vals = {}
for n in "abc":
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/gp_{n}/p_counter_collection.csv")))
    except FileNotFoundError:
        continue
    agg = collections.defaultdict(list)
    for r in rows:
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
rows = list(csv.DictReader(open("gpurun_out/gp_a/p_kernel_trace.csv")))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
dur = sum(d) / len(d)
cyc = vals["GRBM_GUI_ACTIVE"] / 8
print(f"kernel {dur:.1f} us, {cyc:.0f} cycles/XCD -> {cyc/dur/1e3:.2f} GHz")
print(f"MFMA busy {vals['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc*100:.1f} % of kernel cycles ({vals.get('SQ_INSTS_MFMA',0):.0f} MFMAs)")
wc = vals["SQ_WAVE_CYCLES"]
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
          "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"):
    if k in vals:
        print(f"  {k:24s} {vals[k]/wc*100:5.1f} % of wave cycles")
print(f"  wave cycles x4 / (kernel cycles x 256 CUs): {wc*4/cyc/256:.2f} waves resident per CU on average")
for k in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
    if k in vals:
        print(f"  {k:24s} {vals[k]/256/cyc*100:5.1f} % of CU cycles")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"):
    if k in vals:
        print(f"  {k:24s} {vals[k]:.0f}")
