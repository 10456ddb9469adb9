# round 6: the batch behind profiles/r06_* -- one gpurun call:  gpurun --timeout 2400 -- 'bash tools/round6_measurements.sh [tag]'
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06m}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err
python bench.py --arch swin_l_1dl --no-cpu-baseline > $O/bench_swin_l.json 2> $O/bench_swin_l.err
python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 /tmp/p1
RBA_BENCH_NO_EXTRA_LEGS=1 rocprofv3 --kernel-trace --stats -d /tmp/p3 -o bench -- python $R/bench.py --no-cpu-baseline --sustain 0 > $O/prof3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p3 -name "*.db" | head -1) > $O/bench_kernel_trace.md
RBA_BENCH_NO_EXTRA_LEGS=1 rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --sustain 0 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_streams1_kernel_trace.md
python $R/tools/prof_summary.py --sequence $(find /tmp/p1 -name "*.db" | head -1) > $O/step_sequence.md
rm -rf /tmp/p3 /tmp/p1
# matrix-pipe utilisation, Swin-B (configs[1]) AND Swin-L (configs[3]), counters in their own pass (kernel trace only)
for ARCH in swin_b_1dl swin_l_1dl; do
  timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/mfma_$ARCH -o p -- python $R/bench.py --arch $ARCH --steps 3 --warmup 2 --streams 1 --no-cpu-baseline --sustain 0 > $O/mfma_$ARCH.log 2>&1
  f=$(find /tmp/mfma_$ARCH -name "*counter_collection.csv" | head -1); mkdir -p /tmp/mf_$ARCH; cp $f /tmp/mf_$ARCH/p_counter_collection.csv
  python $R/tools/pmc_mfma_parse.py /tmp/mf_$ARCH > $O/mfma_util_$ARCH.md
  rm -rf /tmp/mfma_$ARCH /tmp/mf_$ARCH
done
# K1 HBM traffic on THIS build (FETCH_SIZE / WRITE_SIZE, one counter per pass) -> profiles/k1_pmc.json
cd $R
bash tools/pmc_k1_traffic.sh > $O/pmc_k1.log 2>&1
python tools/pmc_k1_traffic.py > $O/k1_pmc.json 2> $O/k1_pmc.err
cp profiles/k1_pmc.json $O/k1_pmc_profiles_copy.json
rm -rf gpurun_out/k1_FETCH_SIZE gpurun_out/k1_WRITE_SIZE
python tools/stage_times.py > $O/stage_times.md 2> $O/stage_times.err
python tools/evaluator_bench.py 288 > $O/evaluator_288.json 2> $O/evaluator_288.err

python - <<'PY'
import json,glob,os,sys
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); tag=sys.argv[1] if len(sys.argv)>1 else "r06m"
for f in sorted(glob.glob(R+"/gpurun_out/*/bench_*.json")):
    if "/r06" not in f: continue
    try:
        d=json.load(open(f)); print(f.split("/")[-2], f.split("/")[-1], round(d["value"],1), "up4", d.get("value_up4"), "bf16x6", d.get("value_bf16x6"), "flips", d.get("argmax_flips_c2"), d.get("single_stream",{}).get("images_per_s"), round(d["roofline"]["frac"],3), round(d.get("roofline_gemm",{}).get("frac",0),3), d.get("sustained",{}).get("images_per_s"))
    except Exception as e: print(f, "ERR", e)
PY
grep -n "steady-state" -A3 $O/bench_streams1_kernel_trace.md | head; tail -5 $O/mfma_util_swin_l_1dl.md; cut -c1-300 $O/evaluator_288.json | head -3; tail -3 $O/*.err | cut -c1-300
