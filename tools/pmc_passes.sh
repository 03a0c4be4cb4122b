#!/bin/bash
# rocprofv3 counter passes for bench.py (each counter group in its own run, with --kernel-trace only).
# usage on the GPU box: bash tools/pmc_passes.sh   -> gpurun_out/prof/<pass>/.../*counter_collection.csv
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline}
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/prof/$name -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE && run write WRITE_SIZE \
 && run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
 && run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES \
 && run tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
find $R/gpurun_out/prof -name "*counter_collection.csv" | head
