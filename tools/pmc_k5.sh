# PMC passes for K5 on the Swin-B stage-3 shape (separate passes, kernel trace only): bash tools/pmc_k5.sh; python tools/pmc_parse.py swin_window_attn_h3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { # name counters...
  n=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/gp_$n -o p -- python $R/tools/k5_one.py 64 128 16 0 5 > $R/gpurun_out/gp_$n.log 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run c SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
cd $R && python tools/pmc_parse.py swin_window_attn_h3
