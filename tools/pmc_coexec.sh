cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/coexec -o p -- python $R/tools/gemm_one.py 8192 2048 512 3 > $R/gpurun_out/coexec.log 2>&1
rm -f $R/gpurun_out/coexec/p_kernel_trace.csv; tail -1 $R/gpurun_out/coexec.log
