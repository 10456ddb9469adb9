# the batch behind profiles/r04_*: one gpurun call.   bash tools/round4_measurements.sh
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err
python bench.py --arch swin_l_1dl --no-cpu-baseline > $O/bench_swin_l.json 2> $O/bench_swin_l.err
python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
RBA_K6_RS=1 python bench.py --no-cpu-baseline --sustain 0 > $O/bench_default_rs_off.json 2>> $O/err.txt
RBA_K6_RS=0 python bench.py --no-cpu-baseline --sustain 0 > $O/bench_default_rule_only.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o bench -- python $R/bench.py --no-cpu-baseline --sustain 0 > $O/prof3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p3 -name "*.db" | head -1) > $O/bench_kernel_trace.md
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --sustain 0 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_streams1_kernel_trace.md
rm -rf /tmp/p3 /tmp/p1
for ARCH in swin_b_1dl swin_l_1dl; do
  timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/mfma_$ARCH -o p -- python $R/bench.py --arch $ARCH --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --sustain 0 > $O/mfma_$ARCH.log 2>&1
  f=$(find /tmp/mfma_$ARCH -name "*counter_collection.csv" | head -1); mkdir -p /tmp/mf_$ARCH; cp $f /tmp/mf_$ARCH/p_counter_collection.csv
  python $R/tools/pmc_mfma_parse.py /tmp/mf_$ARCH > $O/mfma_util_$ARCH.md
  rm -rf /tmp/mfma_$ARCH /tmp/mf_$ARCH
done
cd $R
python tools/k5_sweep.py > $O/k5_sweep.txt 2>&1
python tools/k1_up4_ab.py 2>&1 | grep -v amdgpu.ids > $O/k1_up4_ab.txt
python tools/evaluator_bench.py 96 > $O/evaluator.json 2> $O/evaluator.err
python tools/evaluator_bench.py 288 > $O/evaluator_288.json 2> $O/evaluator_288.err
timeout 600 python tools/rescore_soak.py 300 > $O/rescore_soak.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r04f/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"), round(d["roofline"]["frac"],3), round(d.get("roofline_gemm",{}).get("frac",0),3), d.get("sustained",{}).get("images_per_s"), d.get("sustained",{}).get("sclk_mhz_mean"))
    except Exception as e: print(f, "ERR", e)
PY
grep -n "steady-state" -A3 $O/bench_streams1_kernel_trace.md; tail -5 $O/mfma_util_swin_b_1dl.md; tail -3 $O/k5_sweep.txt; cut -c1-600 $O/evaluator.json | head -5; cat $O/rescore_soak.txt | cut -c1-300
