set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4n; mkdir -p $O
cd $R
for i in 1 2 3 4; do timeout 900 python -m pytest tests -m gpu -q > $O/tests_$i.txt 2>&1; tail -2 $O/tests_$i.txt; done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
