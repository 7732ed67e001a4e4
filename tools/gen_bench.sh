#!/bin/bash
# the general fused stage on the three workloads of BASELINE.json configs[3]'s element classes: stage time, kernel times, stamps
for w in tets prisms mixed; do
  python bench.py --workload $w --steps 4 --warmup 1 --reps 3 --no-cpu --opt flux_stamps=1 "$@" 2> gpurun_out/gen_$w.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$w', 'stage %.4f ms' % d['ms_per_rk_stage'], '%.2f G' % (d['value']/1e9), {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"
  grep cycles gpurun_out/gen_$w.err | head -8
done
