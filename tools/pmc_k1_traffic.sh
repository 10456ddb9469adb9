# HBM traffic of the product K1 kernel from PMC counters, one counter per pass (gpurun refuses --pmc with other trace domains):
#   bash tools/pmc_k1_traffic.sh [H W] ; python tools/pmc_k1_traffic.py [H W]        (default 1024 2048)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
HWARGS=""; TAG=""
if [ -n "$1" ]; then HWARGS="--hw $1 $2"; TAG="_$1x$2"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/k1${TAG}_$c
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/k1${TAG}_$c -o p -- python $R/tools/k1_sweep.py 121 $HWARGS > $R/gpurun_out/k1${TAG}_$c.log 2>&1
  tail -1 $R/gpurun_out/k1${TAG}_$c.log | cut -c1-120
done
