# the batch behind profiles/r03_*: one gpurun call.   bash tools/round3_measurements.sh
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err
python bench.py --arch swin_l_1dl --no-cpu-baseline > $O/bench_swin_l.json 2> $O/bench_swin_l.err
python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o bench -- python $R/bench.py --no-cpu-baseline > $O/prof3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p3 -name "*.db" | head -1) > $O/bench_kernel_trace.md
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 10 --warmup 3 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/bench_streams1_kernel_trace.md
rm -rf /tmp/p3 /tmp/p1
# matrix-pipe utilisation per kernel (counters in their own pass)
for ARCH in swin_b_1dl swin_l_1dl; do
  timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d /tmp/mfma_$ARCH -o p -- python $R/bench.py --arch $ARCH --steps 3 --warmup 2 --streams 1 --no-cpu-baseline > $O/mfma_$ARCH.log 2>&1
  f=$(find /tmp/mfma_$ARCH -name "*counter_collection.csv" | head -1); mkdir -p /tmp/mf_$ARCH; cp $f /tmp/mf_$ARCH/p_counter_collection.csv
  python $R/tools/pmc_mfma_parse.py /tmp/mf_$ARCH > $O/mfma_util_$ARCH.md
  rm -rf /tmp/mfma_$ARCH /tmp/mf_$ARCH
done
cd $R
bash tools/pmc_k2.sh fused; python tools/pmc_k2_parse.py fused > $O/k2_pmc_final.txt; rm -rf gpurun_out/k2pmc_*
python tools/k2_ab.py > $O/k2_ab.txt 2>&1
python tools/k5_sweep.py > $O/k5_sweep.txt 2>&1
for S in swin_b swin_l c5; do echo "== $S"; python tools/k6_h3q_ab.py $S 30 2>&1 | grep -v amdgpu.ids; done > $O/k6_h3q.txt
python tools/k4_sweep.py 2>&1 | grep -v amdgpu.ids > $O/k4_sweep.txt
python tools/skinny_ab.py 2>&1 | grep -v amdgpu.ids > $O/skinny_ab.txt
python tools/k1_up4_ab.py 2>&1 | grep -v amdgpu.ids > $O/k1_up4_ab.txt
python tools/mask_features_ab.py 2>&1 | grep -v amdgpu.ids > $O/mask_features_ab.txt
python tools/evaluator_bench.py 96 > $O/evaluator.json 2> $O/evaluator.err
cd /tmp; rm -rf /tmp/pc5
rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o bench -- python $R/bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline --streams 1 --steps 10 --warmup 3 > $O/prof_c5.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/pc5 -name "*.db" | head -1) > $O/c5_kernel_trace.md
rm -rf /tmp/pc5
