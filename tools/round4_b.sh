# Round 4, second GPU call: unpacked-fp32 build vs product build on the K6 sub-tile kernel (deferred epilogue), bench with the new fields, rescore soak
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4b; mkdir -p $O
cd $R
python tools/k6_h3q_ab.py swin_b 30 2>&1 | grep -v amdgpu.ids > $O/k6_h3q_packed.txt
RBA_HIP_LIB=$R/tools/ab/librba_hip_nopk.so python tools/k6_h3q_ab.py swin_b 30 2>&1 | grep -v amdgpu.ids > $O/k6_h3q_unpacked.txt
timeout 900 python tools/rescore_soak.py 300 > $O/rescore_soak.json 2> $O/rescore_soak.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err
cat $O/k6_h3q_packed.txt $O/k6_h3q_unpacked.txt; cat $O/rescore_soak.json | cut -c1-1200; tail -3 $O/rescore_soak.err
python - <<'PY'
import json
for f in ("bench_default","bench_streams1"):
    try:
        d=json.load(open(f"/root/repo/gpurun_out/r4b/{f}.json"))
        print(f, d["value"], d["dtype"], d.get("sustained"), d["roofline"]["frac"], d.get("roofline_gemm",{}).get("frac"), d.get("single_stream"), d["wall_time_s"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $O/bench_default.err
