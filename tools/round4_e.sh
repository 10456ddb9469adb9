set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
python tools/token_linear_ab.py 2>&1 | grep -v amdgpu.ids > $O/token_linear_ab.txt; cat $O/token_linear_ab.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -8 $O/tests.txt
python bench.py --streams 1 --no-cpu-baseline --sustain 0 > $O/bench_s1.json 2> $O/bench_s1.err
python bench.py --no-cpu-baseline --sustain 0 > $O/bench_s3.json 2> $O/bench_s3.err
python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline --sustain 0 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4e/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --sustain 0 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_streams1_kernel_trace.md
grep -n "steady-state" -A3 $O/bench_streams1_kernel_trace.md; grep "at::native\|Cijk\|token_linear\|copyBuffer\|skinny" $O/bench_streams1_kernel_trace.md | cut -c1-160 | tail -24
