cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/k1c -o p -- python $R/tools/k1_sweep.py 2 > $R/gpurun_out/k1c.log 2>&1
tail -2 $R/gpurun_out/k1c.log
