set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4r${K6RS:-0}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps1
RBA_K6_RS=${K6RS:-0} rocprofv3 --kernel-trace --stats -d /tmp/ps1 -o bench -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 8 --warmup 3 --sustain 0 > $O/prof_s1.log 2>&1
DB=$(find /tmp/ps1 -name "*.db" | head -1)
python $R/tools/prof_summary.py --sequence $DB > $O/s1_sequence.md
python $R/tools/prof_summary.py $DB > $O/s1_kernel_trace.md
head -5 $O/s1_sequence.md
