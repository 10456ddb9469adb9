set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4m; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -5 $O/tests.txt
python tools/mlp_fused_ab.py > $O/mlp_fused_ab.txt 2>&1; tail -4 $O/mlp_fused_ab.txt
python bench.py --no-cpu-baseline --sustain 0 > $O/bench_s3.json 2>> $O/err.txt
python bench.py --streams 1 --no-cpu-baseline --sustain 0 > $O/bench_s1.json 2>> $O/err.txt
python bench.py --arch swin_l_1dl --no-cpu-baseline --sustain 0 > $O/bench_l.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4m/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"), round(d.get("roofline_gemm",{}).get("frac",0),3), round(d.get("roofline_gemm",{}).get("avg_launch_ms",0)*1e3,1))
    except Exception as e: print(f, "ERR", e)
PY
