#!/bin/bash
# (needs the experiments build of the library: LSM2D_EXPERIMENTS=1)
# A/B of the XCD lockstep on BASELINE configs[4] (1000 scans vs a 1M-point map): kernel ms, in-kernel clock and the L2's hit / miss / fabric-read counters per window.
# usage on the GPU box: bash tools/xcd_lockstep_ab.sh <tag> [map points] [windows...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-xcd}; n=${2:-1000000}; shift; shift; W=${@:-0 1 2 3 5}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
: > $O/xcd_lockstep_ab.jsonl
for w in $W; do
  LSM2D_BENCH_OPTIONS=xcd_lockstep=$w timeout -k 10 60 LSM2D_EXPERIMENTS=1 python bench.py --map-points $n --steps 5 --warmup 2 --spinup-s 0.2 --no-cpu-baseline > $O/line_w$w.json 2>> $O/xcd_lockstep_ab.err || { echo "window $w failed or timed out: stopping"; exit 1; }
  tail -1 $O/line_w$w.json >> $O/xcd_lockstep_ab.jsonl
done
python - <<PY
import json
for l in open("$O/xcd_lockstep_ab.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print("window %s: %8.0f /s  kernel %.3f ms  clock %.0f MHz  wg lifetime %s ms  ok=%s" % (d.get("options", {}).get("xcd_lockstep"), d["value"], r["kernel_ms"], r["clock_mhz_in_kernel"] or 0, r["workgroup_lifetime_ms"], d["parity_ok"]))
PY
cd /tmp; export TMPDIR=/tmp
for w in $W; do
  for c in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    LSM2D_BENCH_OPTIONS=xcd_lockstep=$w timeout -k 10 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_w$w/${c// /_} -- python3 $R/bench.py --map-points $n --steps 2 --warmup 1 --spinup-s 0.05 --no-cpu-baseline > "$O/pmc_w$w.${c// /_}.log" 2>&1 || { echo "pmc window $w [$c] failed: stopping"; tail -3 "$O/pmc_w$w.${c// /_}.log"; exit 1; }
  done
  echo "window $w:"; (cd $R && python tools/pmc_summary.py $O/pmc_w$w k_align | tee $O/pmc_k_align_w$w.csv | tr "\n" " "); echo
done
