set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --sustain 0 > $O/b_s3.json 2>> $O/err.txt
for S in 2 3 4; do
  python bench.py --no-cpu-baseline --sustain 0 --streams $S --cu-partition 1 > $O/b_s${S}_part.json 2>> $O/err.txt
  RBA_K6_RS=2 python bench.py --no-cpu-baseline --sustain 0 --streams $S --cu-partition 1 > $O/b_s${S}_part_rs2.json 2>> $O/err.txt
done
tail -5 $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4k/b_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],2), d["config"].get("cu_partition"))
    except Exception as e: print(f, "ERR", e)
PY
