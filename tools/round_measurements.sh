set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err
python bench.py --streams 1 --no-cpu-baseline > $O/r02_bench_streams1.json 2> $O/r02_bench_streams1.err
python bench.py --arch swin_l_1dl --no-cpu-baseline > $O/r02_bench_swin_l.json 2> $O/r02_bench_swin_l.err
python bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline > $O/r02_bench_c5.json 2> $O/r02_bench_c5.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p3 /tmp/p1
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o bench -- python $R/bench.py --no-cpu-baseline > $O/prof3.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p3 -name "*.db" | head -1) > $O/r02c_bench_kernel_trace.md
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 > $O/prof1.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/p1 -name "*.db" | head -1) > $O/r02d_bench_streams1_kernel_trace.md
cd $R
python tools/evaluator_bench.py 96 > $O/r02_evaluator_b.json 2> $O/r02_evaluator_b.err
python tools/gemm_h3_sweep.py swin_b 4004 > $O/r02_k6_sweep_final.txt 2>&1
