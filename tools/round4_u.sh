# round 4 (late): GPU suite on the build with the mask-feature GroupNorm fold and the K-split form; same-box A/B of both on one stream and three
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for rep in 1 2; do
  for cfg in "1 1" "0 1" "1 0" ; do
    set -- $cfg
    for s in 1 3; do
      KSV=$([ "$2" = "1" ] && echo 0 || echo 1)
      RBA_MF_GN_FOLD=$1 RBA_K6_KS=$KSV python bench.py --no-cpu-baseline --sustain 0 --steps 20 --warmup 5 --streams $s > $O/b_f$1_k$2_s${s}_r${rep}.json 2> $O/b_f$1_k$2_s${s}_r${rep}.err
      python - <<PY
import json
j=json.load(open("$O/b_f$1_k$2_s${s}_r${rep}.json"))
print("gn-fold $1 k-split $2 streams $s rep $rep: images/s %.1f  single %s" % (j["value"], j.get("single_stream",{}).get("images_per_s")))
PY
    done
  done
done
python tools/k6_ks2_ab.py 2>&1 | tail -6
