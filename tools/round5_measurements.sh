# round 5: the batch behind profiles/r05_* -- one gpurun call.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err
python bench.py --arch swin_l_1dl --no-cpu-baseline > $O/bench_swin_l.json 2> $O/bench_swin_l.err
python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o bench -- python $R/bench.py --no-cpu-baseline --sustain 0 > $O/prof3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p3 -name "*.db" | head -1) > $O/bench_kernel_trace.md
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --sustain 0 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_streams1_kernel_trace.md
python $R/tools/prof_summary.py --sequence $(find /tmp/p1 -name "*.db" | head -1) > $O/step_sequence.md
rm -rf /tmp/p3 /tmp/p1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/mfma_b -o p -- python $R/bench.py --arch swin_b_1dl --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --sustain 0 > $O/mfma_b.log 2>&1
f=$(find /tmp/mfma_b -name "*counter_collection.csv" | head -1); mkdir -p /tmp/mf_b; cp $f /tmp/mf_b/p_counter_collection.csv
python $R/tools/pmc_mfma_parse.py /tmp/mf_b > $O/mfma_util_swin_b_1dl.md
rm -rf /tmp/mfma_b /tmp/mf_b
cd $R
python tools/stage_times.py > $O/stage_times.md 2> $O/stage_times.err
python tools/evaluator_bench.py 288 > $O/evaluator_288.json 2> $O/evaluator_288.err
python tools/k7_ab.py > $O/k7_ab.txt 2>&1
python tools/k7_timeline.py > $O/k7_timeline.txt 2>&1

python - <<'PY'
import json,glob,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for f in sorted(glob.glob(R+"/gpurun_out/r05m/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"), round(d["roofline"]["frac"],3), round(d.get("roofline_gemm",{}).get("frac",0),3), d.get("sustained",{}).get("images_per_s"), d.get("sustained",{}).get("sclk_mhz_mean"))
    except Exception as e: print(f, "ERR", e)
PY
grep -n "steady-state" -A3 $O/bench_streams1_kernel_trace.md | head; tail -5 $O/mfma_util_swin_b_1dl.md; cut -c1-500 $O/evaluator_288.json | head -3
