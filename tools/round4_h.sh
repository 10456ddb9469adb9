set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "split_linear or conv3x3 or mlp" > $O/tests_k.txt 2>&1; tail -4 $O/tests_k.txt
for RS in 1 0 2; do
  RBA_K6_RS=$RS python bench.py --streams 1 --no-cpu-baseline --sustain 0 > $O/bench_b_s1_rs$RS.json 2> $O/err.txt
  RBA_K6_RS=$RS python bench.py --no-cpu-baseline --sustain 0 > $O/bench_b_s3_rs$RS.json 2>> $O/err.txt
  RBA_K6_RS=$RS python bench.py --arch swin_l_1dl --no-cpu-baseline --sustain 0 > $O/bench_l_s3_rs$RS.json 2>> $O/err.txt
  RBA_K6_RS=$RS python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline --sustain 0 > $O/bench_c5_s3_rs$RS.json 2>> $O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4h/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"), round(d.get("roofline_gemm",{}).get("frac",0),3), round(d.get("roofline_gemm",{}).get("avg_launch_ms",0)*1e3,1))
    except Exception as e: print(f, "ERR", e)
PY
