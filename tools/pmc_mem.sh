# memory-path counters for one K6 v4/v5 launch shape: bash tools/pmc_mem.sh M N K cfg
cd /tmp && export TMPDIR=/tmp
R=/root/repo
M=${1:-8192}; N=${2:-2048}; K=${3:-512}; CFG=${4:-}
run() { n=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/gm_$n -o p -- python $R/tools/gemm_one.py $M $N $K 3 $CFG > $R/gpurun_out/gm_$n.log 2>&1
}
run a TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run c TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run d TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run e FETCH_SIZE
run f TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
