#!/bin/bash
# Dump the gfx950 ISA of the library's kernels and their register / scratch budget (no GPU needed).
# usage: bash tools/isa_dump.sh [extra hipcc flags]   -> /tmp/lsm2d_isa/lsm2d.s, resource summary on stdout
R=$(cd "$(dirname "$0")/.." && pwd); O=/tmp/lsm2d_isa; mkdir -p $O; cd $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared "$@" \
  -I$R/include -I$R/srrg2_laser_slam_2d_amd/csrc -save-temps -o $O/lib.so $R/srrg2_laser_slam_2d_amd/csrc/lsm2d_capi.hip 2> $O/build.err || { cat $O/build.err; exit 1; }
cp lsm2d_capi-hip-amdgcn-amd-amdhsa-gfx950.s lsm2d.s
python3 - <<'PY'
import re
s = open('/tmp/lsm2d_isa/lsm2d.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, '?'])[1]
    print('%-70s vgpr %-4s sgpr %-4s scratch %-5s lds %s' % (name[:70], g('next_free_vgpr'), g('next_free_sgpr'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
for m in re.finditer(r'; Function info:.*?\n(.*?)(?=\n\t\.|\Z)', s, re.S):
    pass
PY
grep -n "NumVgprs\|ScratchSize\|Occupancy\|SpillCount\|; -- Begin function\|sgpr_spill_count\|vgpr_spill_count" lsm2d.s | grep -A6 "k_alignILb1ELb0ELb0" | head -12
