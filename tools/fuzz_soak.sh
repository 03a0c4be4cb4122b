#!/bin/bash
# Soak of the three randomised GPU tests over many seeds: bash tools/fuzz_soak.sh <out.log> [trials] [seed ...]
# Every aligner run of the two aligner fuzz tests is also made with "sum_order" 1 and held BITWISE to the sequential fp32 oracle (round 6); the default order is held
# to the envelope with 64 perturbed starts, and only the alignments named in tests/fuzz_cases.py may lie outside it.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; out=$1; trials=${2:-420}; shift 2
seeds=${@:-"1 2 3 5 7 11 42 99 123 777 2024 5150 31337 65537 17 4711 271828 8675309"}
cd $R; : > $out
for s in $seeds; do
  echo "== seed $s, $trials trials" >> $out
  LSM2D_FUZZ_TRIALS=$trials LSM2D_FUZZ_SEED=$s timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_mapping.py -m gpu -q -s -k "randomised" >> $out 2>&1 || { echo "FAILED seed $s" >> $out; tail -30 $out; exit 1; }
done
grep -c "3 passed" $out
