# round 4 (late): GPU suite on the build with the GroupNorm moments from the FPN GEMM epilogues; same-box A/B (RBA_GN_MOMENTS)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4v; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -12 $O/tests.txt
for rep in 1 2; do
  for m in 1 0; do
    for s in 1 3; do
      RBA_GN_MOMENTS=$m python bench.py --no-cpu-baseline --sustain 0 --steps 20 --warmup 5 --streams $s > $O/b_m${m}_s${s}_r${rep}.json 2> $O/b_m${m}_s${s}_r${rep}.err
      python - <<PY
import json
j=json.load(open("$O/b_m${m}_s${s}_r${rep}.json"))
print("gn-moments $m streams $s rep $rep: images/s %.1f  single %s" % (j["value"], j.get("single_stream",{}).get("images_per_s")))
PY
    done
  done
done
