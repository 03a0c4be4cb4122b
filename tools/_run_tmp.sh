mkdir -p gpurun_out/r04j
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/kd_build_bench.py --map-points 100000 --scans 1000 2>/dev/null | head -4 | cut -c1-250
python tests/bench/track_step_bench.py --steps 2000 > gpurun_out/r04j/track_step_c_abi.json 2>gpurun_out/r04j/track_step.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j/track_step_c_abi.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'ms_per_step_wall' in v: print(k, round(v['ms_per_step_wall'],4), v['status'], v['host_us_in_calls'])
PY
