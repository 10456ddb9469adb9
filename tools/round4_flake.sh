# the intermittent mismatch of test_non_finite_f16x3_score_is_rescored_on_bf16x6 only shows in full-suite runs: repeat the suite, keep the diagnostics
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4flake; mkdir -p $O
cd $R
for i in 1 2 3 4 5; do
  python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_$i.txt 2>&1
  tail -1 $O/tests_$i.txt
  grep -n "evaluator vs eager" $O/tests_$i.txt | head -3
done
