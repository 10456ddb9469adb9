# same-box A/B of the 256 x 128 K6 form's K threshold (RBA_K6_RS_MIN_K): 0 = the tile-count rule alone, 512 = only from K = 512 on
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R
for rep in 1 2; do
  for mk in 0 512; do
    for s in 3 1; do
      RBA_K6_RS_MIN_K=$mk python bench.py --no-cpu-baseline --sustain 0 --steps 20 --warmup 5 --streams $s > $O/b_mk${mk}_s${s}_r${rep}.json 2> $O/b_mk${mk}_s${s}_r${rep}.err
      python - <<PY
import json
j=json.load(open("$O/b_mk${mk}_s${s}_r${rep}.json"))
print("min_k $mk streams $s rep $rep: images/s %.1f  single %s  fc1 probe %s" % (j["value"], j.get("single_stream_images_per_s"), j.get("roofline_gemm",{}).get("us_per_launch")))
PY
    done
  done
done
