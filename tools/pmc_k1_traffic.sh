# HBM traffic of the product K1 kernel from PMC counters, one counter per pass (gpurun refuses --pmc with other trace domains):
#   bash tools/pmc_k1_traffic.sh ; python tools/pmc_k1_traffic.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/k1_$c -o p -- python $R/tools/k1_sweep.py 121 > $R/gpurun_out/k1_$c.log 2>&1
  tail -1 $R/gpurun_out/k1_$c.log | cut -c1-120
done
