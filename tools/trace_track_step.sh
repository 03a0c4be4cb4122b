#!/bin/bash
# GPU timeline of the live tracker's step: rocprofv3 --kernel-trace over the bare C-ABI driver (asynchronous mode, 200 steps),
# then the per-kernel durations and the idle gaps between consecutive kernels of a step.
# usage (GPU box): bash tools/trace_track_step.sh [mode]     -> gpurun_out/track_trace_summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; MODE=${1:-1}
mkdir -p $R/gpurun_out; W=/tmp/tsb_work; rm -rf $W /tmp/tsb_prof
cd $R && python tests/bench/track_step_bench.py --steps 200 --workdir $W > /dev/null 2>&1 || { echo "bench failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tsb_prof -- $(cat $W/cmd_$MODE.txt) > /dev/null 2>&1
python3 - <<'PY' > $R/gpurun_out/track_trace_summary.txt
import csv, glob, collections
f = glob.glob('/tmp/tsb_prof/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(({'k': r['Kernel_Name'].split('(')[0][:40], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])} for r in csv.DictReader(open(f))), key=lambda r: r['s'])
rows = rows[len(rows) // 2:]                       # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[a['k']].append(a['e'] - a['s']); gap[a['k'] + ' -> ' + b['k']].append(b['s'] - a['e'])
print('kernel durations [us]: mean over the second half of the run')
for k, v in dur.items(): print('  %-42s n=%4d mean=%7.2f' % (k, len(v), sum(v) / len(v) / 1e3))
print('gaps between consecutive kernels [us]')
for k, v in gap.items(): print('  %-86s n=%4d mean=%7.2f' % (k, len(v), sum(v) / len(v) / 1e3))
span = (rows[-1]['e'] - rows[0]['s']) / 1e3; busy = sum(r['e'] - r['s'] for r in rows) / 1e3
print('span %.1f us, busy %.1f us (%.0f %%), kernels %d' % (span, busy, 100 * busy / span, len(rows)))
PY
cat $R/gpurun_out/track_trace_summary.txt
