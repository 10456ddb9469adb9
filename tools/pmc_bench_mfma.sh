# Matrix-pipe utilisation per kernel of one bench run:  bash tools/pmc_bench_mfma.sh [arch]   (counters in their own pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
ARCH=${1:-swin_b_1dl}
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/gpurun_out/mfma_$ARCH -o p -- python $R/bench.py --arch $ARCH --steps 3 --warmup 2 --streams 1 --no-cpu-baseline > $R/gpurun_out/mfma_$ARCH.log 2>&1
rm -f $R/gpurun_out/mfma_$ARCH/p_kernel_trace.csv
tail -1 $R/gpurun_out/mfma_$ARCH.log | cut -c1-150
