#!/bin/bash
# PMC counters of k_align for one build variant: bash tools/pmc_variant.sh <tag> "<hipcc flags>"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=$1; flags=$2
cd $R; LSM2D_EXTRA_HIPCC_FLAGS="$flags" python -m srrg2_laser_slam_2d_amd.build --force > /dev/null 2>&1 || exit 1
mkdir -p $R/gpurun_out/prof_$tag; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/prof_$tag/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$tag/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES
cd $R; echo "== $tag [$flags]"; python tools/pmc_summary.py gpurun_out/prof_$tag k_align
