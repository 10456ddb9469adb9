set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4l; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "256x128 or conv3x3" > $O/tests_k.txt 2>&1; tail -3 $O/tests_k.txt
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "evaluate or cli or datasets or dense" > $O/tests_m.txt 2>&1; tail -3 $O/tests_m.txt
python tools/evaluator_bench.py 96 > $O/evaluator.json 2> $O/evaluator.err
python - <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/r4l/evaluator.json"))
for k,v in d.items():
    if isinstance(v,dict) and "images_per_s" in v: print(k, v["images_per_s"], v.get("host_thread"))
    elif isinstance(v,dict) and "images_per_s_scoring_loop" in v: print(k, v["images_per_s_scoring_loop"])
PY
