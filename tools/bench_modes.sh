#!/bin/bash
# Every BASELINE.json configuration that is a bench.py line, one JSON line each -> gpurun_out/<tag>/bench_modes.jsonl
# usage on the GPU box: bash tools/bench_modes.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; tag=${1:-round}; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
: > $O/bench_modes.jsonl
run() { timeout -k 10 400 python bench.py "$@" 2>> $O/bench_modes.err | tail -1 >> $O/bench_modes.jsonl; echo "mode [$*] rc=$?"; }
run                                                                  # configs[1], role A / projective (the headline line)
run --role B --finder nn --cpu-sample 200                            # configs[1], role B / NN
run --role B --finder kdtree --cpu-sample 200                        # configs[1], role B / the reference's own KD-tree (BASELINE's wording: tree over the map, scans as queries)
run --role A --finder nn --max-distance 0.3 --cpu-sample 100 --steps 5      # configs[1], role A / NN
run --role A --finder kdtree --max-distance 0.3 --cpu-sample 100 --steps 5  # configs[1], role A / KD-tree (a tree per scan, every map point a query: the tracker's wiring)
run --role A --finder distmap --max-distance 0.5 --cpu-sample 100 --steps 10       # configs[1], distance-map finder (CorrespondenceFinderNN2D, row f4): a map per scan
run --role B --finder distmap --max-distance 0.5 --cpu-sample 200 --steps 100      # same finder, ONE map over the 100k-point cloud serving every alignment (SURVEY f4's case)
run --map-points 1000000 --cpu-sample 100 --steps 5                  # configs[4]
run --scans 65536 --unique-scans 2048 --cauchy 0.05 --steps 3 --warmup 1 --cpu-sample 1000      # configs[3], one GPU's view (Cauchy 0.05: MULTI.json:957-962)
run --scans 1 --map-points 10000 --steps 2000 --warmup 2000 --cpu-sample 1       # configs[0] (long warm-up: the first ~0.1 s of calls in a process wake the host side up slowly)
python - <<PY
import json
for l in open("$O/bench_modes.jsonl"):
    d = json.loads(l); c = d.get("cpu_baseline") or {}
    print("%-110s | %9.0f /s | kernel %8.3f ms | cpu %s | ok=%s" % (d["config"]["workload"][:110], d["value"], d["roofline"]["kernel_ms"], c.get("value"), d["parity_ok"]))
PY
