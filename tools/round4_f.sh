set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
python tools/token_linear_ab.py 2>&1 | grep -v amdgpu.ids > $O/token_linear_ab.txt; cat $O/token_linear_ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "token_linear or skinny" > $O/tests_k.txt 2>&1; tail -5 $O/tests_k.txt
python bench.py --streams 1 --no-cpu-baseline --sustain 0 > $O/bench_s1.json 2> $O/bench_s1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4f/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"))
    except Exception as e: print(f, "ERR", e)
PY
