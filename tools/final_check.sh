#!/bin/bash
# the round's last GPU check: the whole GPU suite, the default bench line (summary of its fields), the smoke entry
python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1; tail -4 gpurun_out/t_full.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_rk_stage"], d["roofline"]["frac"], d["api_path_ms_per_rk_stage"], d["cpu_baseline"]["value"])
for k, v in d["also"].items():
    print(k, v.get("ms_per_rk_stage"), v.get("error"), v.get("leg_wall_s"))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
