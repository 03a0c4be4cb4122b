#!/bin/bash
# Past-L3 size sweep of the big-map configuration: the projective headline kernel at 1M / 4M / 16M / 64M map points -- lane copy 8 / 32 / 128 / 512 MB,
# AoS copy 16 / 64 / 256 / 1024 MB: the last STREAM is twice the 256 MiB Infinity Cache -- kernel time, in-kernel clock, and (second half) the fabric-side
# counters per launch, each counter group in its own rocprofv3 pass.  SIZES="..." overrides the list.
# usage on the GPU box: bash tools/size_sweep.sh <tag> [extra bench flags]   -> gpurun_out/<tag>/size_sweep.jsonl, pmc_<size>/...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-sweep}; shift; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
: > $O/size_sweep.jsonl
for n in ${SIZES:-1000000 4000000 16000000 64000000}; do
  timeout -k 10 400 python bench.py --map-points $n --steps 3 --warmup 1 --spinup-s 0.2 --no-cpu-baseline --no-also "$@" 2>> $O/size_sweep.err | tail -1 >> $O/size_sweep.jsonl || { echo "size $n failed"; exit 1; }
  echo "size $n done"
done
cd /tmp; export TMPDIR=/tmp
for n in ${SIZES:-1000000 4000000 16000000 64000000}; do
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n/${c// /_} -- python3 $R/bench.py --map-points $n --steps 2 --warmup 1 --spinup-s 0.05 --no-cpu-baseline "$@" > "$O/pmc_$n.${c// /_}.log" 2>&1 || { echo "pmc $n [$c] failed (counter group unavailable?)"; tail -3 "$O/pmc_$n.${c// /_}.log"; }
  done
  (cd $R && python tools/pmc_summary.py $O/pmc_$n k_align > $O/pmc_k_align_$n.csv; cat $O/pmc_k_align_$n.csv | cut -c1-400)
done
