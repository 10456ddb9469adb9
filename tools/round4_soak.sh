# Round 4, first GPU call: the 16-bit-destination hazard probe, the K1 soak on the round-3 build and on this build, the GPU suite, the bench.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
timeout 120 tools/micro/bin/mix_hazard > $O/mix_hazard.txt 2>&1
RBA_HIP_LIB=$R/tools/ab/librba_hip_r3.so timeout 600 python tools/k1_soak.py 6000 200 > $O/soak_r3.json 2> $O/soak_r3.err
timeout 600 python tools/k1_soak.py 6000 200 > $O/soak_r4.json 2> $O/soak_r4.err
RBA_HIP_LIB=$R/tools/ab/librba_hip_r3.so python tools/k1_up4_ab.py 2>&1 | grep -v amdgpu.ids > $O/k1_up4_ab_r3.txt
python tools/k1_up4_ab.py 2>&1 | grep -v amdgpu.ids > $O/k1_up4_ab_r4.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/tests.txt; cat $O/mix_hazard.txt; cat $O/soak_r3.json $O/soak_r4.json | cut -c1-1500; cat $O/k1_up4_ab_r3.txt $O/k1_up4_ab_r4.txt
