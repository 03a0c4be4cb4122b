#!/bin/bash
# Counters of the live tracker's step (bare C-ABI driver, asynchronous mode): three rocprofv3 --pmc passes over tests/cpp/track_step_bench.cpp,
# per-launch means for k_align_pair, k_merge_multi and k_clip_small -> gpurun_out/<tag>/pmc_track_step.csv
# usage (GPU box): bash tools/pmc_track_step.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-round}; O=$R/gpurun_out/$tag; mkdir -p $O
W=/tmp/tsb_pmc_work; rm -rf $W /tmp/tsb_pmc
cd $R && python tests/bench/track_step_bench.py --steps 200 --workdir $W > /dev/null 2>&1 || { echo "bench failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32" "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/tsb_pmc/${c// /_} -- $(cat $W/cmd_1.txt) > /dev/null 2>&1 || { echo "pass failed: $c"; exit 1; }
done
cd $R
for k in k_align_pair k_merge_multi k_clip_small; do echo "== $k"; python tools/pmc_summary.py /tmp/tsb_pmc $k; done > $O/pmc_track_step.csv
cat $O/pmc_track_step.csv
