#!/bin/bash
# Full evidence pass for bench.py on the GPU box: kernel-trace stats + PMC passes + plain bench line.
# usage: bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>/...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-round}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R && python bench.py --steps 100 --no-also > $O/bench_pre.json 2> $O/bench.err; tail -c 300 $O/bench_pre.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --no-cpu-baseline --no-also > $O/trace.log 2>&1; echo "trace rc=$?"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $O/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE && run write WRITE_SIZE \
 && run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU \
 && run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES \
 && run sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_ATOMIC SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VMEM \
 && run tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
cd $R; python tools/pmc_summary.py $O k_align > $O/pmc_k_align.csv; cat $O/pmc_k_align.csv; cat $O/trace/*/*kernel_stats.csv | head -5
# the stream's own issue rate (register-resident points, no loads / barriers / bin walk): the second yardstick of bench.py's roofline block
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -Wno-unused-value -Isrrg2_laser_slam_2d_amd/csrc -Iinclude -o /tmp/valu_probe tools/valu_issue_probe.hip 2>/dev/null \
  && timeout -k 5 120 /tmp/valu_probe > $O/valu_issue_probe.txt; tail -7 $O/valu_issue_probe.txt | cut -c1-160
timeout -k 5 100 python tools/occupancy_probe.py > $O/occupancy_probe.txt 2>&1; tail -3 $O/occupancy_probe.txt
# counters -> profiles/counters.json (hash-stamped), then the bench line that reads them back
python tools/write_counters.py $O --tag $tag && cp profiles/counters.json $O/counters.json && python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
# the command line the round-end driver used in round 1 (20 timed steps after 5 warm-up steps)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench.err; cut -c1-330 $O/bench_driver_cmd.json
# the rows either side of the path (preprocessor batch, finder reset() structures, clip / merge): wall times + per-kernel durations
cd /tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rows_trace -- python3 $R/tests/bench/rows_bench.py > $O/rows.jsonl 2> $O/rows.err; echo "rows rc=$?"; cd $R
cat $O/rows_trace/*/*kernel_stats.csv | head -12
