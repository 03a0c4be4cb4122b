#!/bin/bash
# Evidence pass (rounds 5-6) on the GPU box, in this order (each part appends to gpurun_out/<tag>/): the GPU test suite; the headline's profile round (trace, PMC, probes,
# counters.json, bench lines incl. the driver's command); the other modes' counters and lines; the size sweep past the Infinity Cache; the streamed pipeline with its
# kernel trace.  usage: bash tools/evidence_round.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-r05}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
(timeout -k 10 700 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log)
(LSM2D_EXPERIMENTS=1 timeout -k 10 700 python -m pytest tests -m gpu -q > $O/gpu_tests_experiments_build.log 2>&1; echo "gpu tests (experiments build) rc=$?"; tail -3 $O/gpu_tests_experiments_build.log)
bash tools/profile_round.sh $tag > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log | cut -c1-300
bash tools/pmc_modes.sh $tag > $O/pmc_modes.log 2>&1; tail -3 $O/pmc_modes.log | cut -c1-200
python bench.py > $O/bench_also.json 2> $O/bench_also.err; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench_also.err
bash tools/bench_modes.sh $tag > $O/bench_modes.log 2>&1; tail -12 $O/bench_modes.log | cut -c1-220
bash tools/size_sweep.sh ${tag}_sweep > $O/size_sweep.log 2>&1; tail -4 $O/size_sweep.log | cut -c1-300
timeout -k 10 200 python bench.py --stream --steps 250 --no-cpu-baseline > $O/stream.json 2> $O/stream.err; cut -c1-400 $O/stream.json
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream_trace -- python3 $R/bench.py --stream --steps 60 --spinup-s 0.1 --no-cpu-baseline > $O/stream_trace.log 2>&1; echo "stream trace rc=$?"
cd $R; python tools/stream_overlap.py $O/stream_trace > $O/stream_overlap.txt 2>&1; cat $O/stream_overlap.txt
