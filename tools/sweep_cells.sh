#!/bin/bash
# per-kernel times of the split3 stage at several box sizes (plane strides that are / are not multiples of large powers of two)
for n in 31 32 33 36; do
  python bench.py --cells $n --steps 10 --warmup 2 --reps 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['roofline']['kernels_ms']; ne=d['config']['n_eles_per_gpu']
print($n, ne, 'stage %.4f'%d['ms_per_rk_stage'], {a:round(b/ne*32768,4) for a,b in k.items()}, 'stage/ele*32768 %.4f'%(d['ms_per_rk_stage']/ne*32768))"
done
