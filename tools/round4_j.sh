set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "hint or bench or evaluate" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for A in "swin_b_1dl 1024 2048 b" "swin_l_1dl 1024 2048 l" "swin_b_9dl 720 1280 c5"; do
  set -- $A
  python bench.py --arch $1 --height $2 --width $3 --no-cpu-baseline --sustain 0 > $O/bench_$4_hint.json 2>> $O/err.txt
  RBA_K6_RS=0 python bench.py --arch $1 --height $2 --width $3 --no-cpu-baseline --sustain 0 > $O/bench_$4_rule.json 2>> $O/err.txt
done
python tools/evaluator_bench.py 96 > $O/evaluator.json 2> $O/evaluator.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4j/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"))
    except Exception as e: print(f, "ERR", e)
d=json.load(open("/root/repo/gpurun_out/r4j/evaluator.json"))
for k,v in d.items():
    if isinstance(v,dict) and "images_per_s" in v: print(k, v["images_per_s"], v.get("host_thread"))
    elif isinstance(v,dict) and "images_per_s_scoring_loop" in v: print(k, v["images_per_s_scoring_loop"])
PY
