set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4o; mkdir -p $O
cd $R
for i in 1 2; do
  RBA_HIP_LIB=$R/tools/ab/librba_hip_k5old.so python tools/k5_sweep.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/scaled P   (before): /' >> $O/k5_ab.txt
  python tools/k5_sweep.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/^/un-scaled P (now):   /' >> $O/k5_ab.txt
done
cat $O/k5_ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "k5 or swin" > $O/tests_k.txt 2>&1; tail -3 $O/tests_k.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x > $O/tests_m.txt 2>&1; tail -3 $O/tests_m.txt
python bench.py --no-cpu-baseline --sustain 0 > $O/bench_s3.json 2>> $O/err.txt
python bench.py --streams 1 --no-cpu-baseline --sustain 0 > $O/bench_s1.json 2>> $O/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r4o/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d.get("single_stream",{}).get("images_per_s"))
PY
