#!/bin/bash
# A/B kernel build variants on the GPU box: for each flag set rebuild the library and run bench.py
# usage: bash tools/variant_bench.sh "" "-DLSM2D_ALIGN_MIN_WAVES=6" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
for flags in "$@"; do
  LSM2D_EXTRA_HIPCC_FLAGS="$flags" python -m srrg2_laser_slam_2d_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps ${STEPS:-40} ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=[%s] rep=%s value=%.0f align/s kernel_ms=%.4f clock_mhz=%s wg_ms=%s parity_ok=%s err=%.2e' % ('$flags', '$rep', d['value'], d['roofline']['kernel_ms'], d['roofline'].get('clock_mhz_in_kernel'), d['roofline'].get('workgroup_lifetime_ms'), d['parity_ok'], d['max_pose_err_m']))"
  done
done | tee -a gpurun_out/variants.log
python -m srrg2_laser_slam_2d_amd.build --force > /dev/null 2>&1
