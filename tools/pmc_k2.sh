# counters of K2 at BASELINE C5's encoder shape (19 320 queries x 3 levels): bash tools/pmc_k2.sh [fused|fwd|generic]
# one counter group per pass, kernel trace only (no other trace domains); summary: python tools/pmc_k2_parse.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
F=${1:-fused}
run() { n=$1; shift
  timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/k2pmc_${F}_$n -o p -- python $R/tools/k2_one.py $F 5 > $R/gpurun_out/k2pmc_${F}_$n.log 2>&1
}
run a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run c TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run d FETCH_SIZE
run e WRITE_SIZE
run f SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run g SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run h TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
