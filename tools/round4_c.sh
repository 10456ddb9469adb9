# Round 4, third GPU call: the row-complete token Linear -- kernel tests, model parity, A/B against the library path
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "token_linear" > $O/tests_token.txt 2>&1; tail -15 $O/tests_token.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x > $O/tests_model.txt 2>&1; tail -15 $O/tests_model.txt
for TL in 0 1; do
  RBA_TOKEN_LINEAR=$TL python bench.py --streams 1 --no-cpu-baseline --sustain 0 > $O/bench_s1_tl$TL.json 2> $O/bench_s1_tl$TL.err
  RBA_TOKEN_LINEAR=$TL python bench.py --no-cpu-baseline --sustain 0 > $O/bench_s3_tl$TL.json 2> $O/bench_s3_tl$TL.err
  RBA_TOKEN_LINEAR=$TL python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline --sustain 0 > $O/bench_c5_tl$TL.json 2> $O/bench_c5_tl$TL.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4c/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --sustain 0 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_streams1_kernel_trace.md
grep -n "steady-state" -A3 $O/bench_streams1_kernel_trace.md; grep -c "Cijk" $O/bench_streams1_kernel_trace.md; grep "at::native\|Cijk\|token_linear\|copyBuffer" $O/bench_streams1_kernel_trace.md | cut -c1-160 | tail -30
