set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4p; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "token_linear" > $O/tests_k.txt 2>&1; tail -3 $O/tests_k.txt
python tools/token_linear_ab.py 2>&1 | grep -v amdgpu.ids > $O/token_linear_ab.txt; cat $O/token_linear_ab.txt
