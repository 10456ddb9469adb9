set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for S in 2 3 4 5 6; do
  python bench.py --streams $S --no-cpu-baseline --sustain 0 --steps 20 > $O/bench_streams$S.json 2>> $O/err.txt
done
python bench.py --arch swin_l_1dl --no-cpu-baseline --sustain 0 > $O/bench_l.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4i/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), d.get("single_stream",{}).get("images_per_s"), round(d.get("roofline_gemm",{}).get("frac",0),3), round(d.get("roofline_gemm",{}).get("avg_launch_ms",0)*1e3,1))
    except Exception as e: print(f, "ERR", e)
PY
