set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc5
rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o bench -- python $R/bench.py --arch swin_b_9dl --height 720 --width 1280 --no-cpu-baseline --streams 1 --steps 10 --warmup 3 --sustain 0 > $O/prof_c5.log 2>&1
python $R/tools/prof_summary.py $(find /tmp/pc5 -name "*.db" | head -1) > $O/c5_kernel_trace.md
awk '/steady-state/{f=1} f' $O/c5_kernel_trace.md | head -48 | cut -c1-175
