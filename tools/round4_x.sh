#!/bin/bash
# Same-box A/B of K5's register budget in the network: bench (3 streams, graphs) and single stream, alternating, + the K5 tests.
mkdir -p gpurun_out/s4e
python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "k5" 2>&1 | tail -2 | tee gpurun_out/s4e/k5_tests.txt
for rep in 1 2; do
  for w in ${WPES:-5 6}; do
    RBA_K5_WPE=$w timeout 150 python bench.py --no-cpu-baseline --sustain 0 --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wpe', $w, 'default', round(d['value'],1), 'single', {k: round(v,1) for k,v in d['single_stream']['images_per_s'].items()})" | tee -a gpurun_out/s4e/bench_ab.txt
  done
done
