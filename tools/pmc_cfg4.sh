#!/bin/bash
# HBM traffic of k_align on BASELINE configs[4] (1000 scans vs a 1M-point map): FETCH_SIZE / WRITE_SIZE / TCC hit-miss, each in its own pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/cfg4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py --map-points 1000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE && run write WRITE_SIZE && run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
cd $R; python tools/pmc_summary.py $O k_align | tee $O/pmc_k_align_cfg4.csv
