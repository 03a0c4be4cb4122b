#!/bin/bash
# Counters for the OTHER bench.py modes (finders / roles / map sizes): three rocprofv3 --pmc passes each (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU),
# merged into profiles/counters.json under the mode's own key, so that every line of tools/bench_modes.sh carries a measured roofline.
# usage on the GPU box, AFTER tools/profile_round.sh <tag> (which writes the headline's entry) and in the SAME gpurun call -- or with that call's
# profiles/counters.json copied into the tree first (a file stamped with another source hash is started afresh): bash tools/pmc_modes.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-round}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
mode() {      # name, write_counters flags, bench flags
  name=$1; wc=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$name/${c// /_} -- python3 $R/bench.py --steps 3 --warmup 1 --spinup-s 0.05 --no-cpu-baseline "$@" > "$O/pmc_$name.${c// /_}.log" 2>&1 || { echo "$name $c failed"; return 1; }
  done
  (cd $R && python tools/write_counters.py $O/pmc_$name $wc --tag $tag | cut -c1-200)
}
mode nnB   "--role B --finder nn"       --role B --finder nn \
&& mode kdB   "--role B --finder kdtree"   --role B --finder kdtree \
&& mode kdA   "--role A --finder kdtree"   --role A --finder kdtree --max-distance 0.3 \
&& mode nnA   "--role A --finder nn"       --role A --finder nn --max-distance 0.3 \
&& mode distA "--role A --finder distmap"  --role A --finder distmap --max-distance 0.5 \
&& mode distB "--role B --finder distmap"  --role B --finder distmap --max-distance 0.5 \
&& mode map1M "--map-points 1000000"       --map-points 1000000 \
&& mode cfg3  "--scans 65536 --unique-scans 2048 --cauchy 0.05" --scans 65536 --unique-scans 2048 --cauchy 0.05
cp $R/profiles/counters.json $O/counters.json
