set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
for S in swin_b swin_l; do echo "== $S"; timeout 600 python tools/k6_rs2_ab.py $S 2>&1 | grep -v amdgpu.ids; done > $O/k6_rs2.txt; cat $O/k6_rs2.txt
