#!/bin/bash
# Diagnostic PMC passes for one bench.py mode: each argument group before "--" is ONE rocprofv3 --pmc pass (a space-separated counter list in
# quotes, at most what the blocks' slots allow); everything after "--" goes to bench.py.  Means per k_align launch -> gpurun_out/<tag>/pmc_probe_<name>.csv
# usage on the GPU box: bash tools/pmc_probe.sh <tag> <name> "SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES TA_BUSY" -- --role B --finder kdtree
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; name=$2; shift 2
O=$R/gpurun_out/$tag; mkdir -p $O
passes=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do passes+=("$1"); shift; done; shift
cd /tmp; export TMPDIR=/tmp
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmcp_$name/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --spinup-s 0.05 --no-cpu-baseline "$@" > $O/pmcp_$name.pass$i.log 2>&1 || echo "pass $i ($p) failed: $(tail -2 $O/pmcp_$name.pass$i.log)"
done
python3 $R/tools/pmc_summary.py $O/pmcp_$name k_align > $O/pmc_probe_$name.csv
cat $O/pmc_probe_$name.csv
rm -rf $O/pmcp_$name
