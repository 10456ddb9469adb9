#!/bin/bash
# rocprofv3 kernel-trace of one command; prints calls / average / min duration (us) of every kernel whose name matches $FILTER.
#   FILTER=h3_kernel bash tools/rocprof_kernels.sh python tools/gemm_h3_sweep.py swin_b 4 s3_fc1
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rk_prof
rocprofv3 --kernel-trace --stats -d /tmp/rk_prof -o rk --output-format csv -- "$@" > /tmp/rk_prof.log 2>&1
f=$(find /tmp/rk_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, os, sys
flt = os.environ.get("FILTER", "")
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if flt in n:
        short = n.split("(")[0] if not n.startswith("void (anonymous") else "(anon)::" + n.split("::", 1)[1].split("(")[0]
        print(f"{int(r['Calls']):5d} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f} us  {short[:110]}")
PY
