#!/bin/bash
# For each flag set: rebuild, one bench line, then the per-workgroup lifetime probe (tools/occupancy_probe.py).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
for flags in "$@"; do
  LSM2D_EXTRA_HIPCC_FLAGS="$flags" python -m srrg2_laser_slam_2d_amd.build --force > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "=== flags=[$flags]"
  python bench.py --no-cpu-baseline --steps ${STEPS:-40} ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value=%.0f align/s kernel_ms=%.4f clock_mhz=%s parity_ok=%s err=%.2e' % (d['value'], d['roofline']['kernel_ms'], d['roofline'].get('clock_mhz_in_kernel'), d['parity_ok'], d['max_pose_err_m']))"
  timeout -k 10 120 python tools/occupancy_probe.py 2>&1 | grep -v amdgpu.ids
done | tee -a gpurun_out/variant_probe.log
python -m srrg2_laser_slam_2d_amd.build --force > /dev/null 2>&1
