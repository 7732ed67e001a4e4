#!/bin/bash
# phase stamps of workgroup 0 of the flux kernel at a few iterations (early / steady state / late)
for it in 2 4 6; do
  echo "== iteration $it"
  python bench.py --steps 4 --warmup 1 --reps 1 --no-cpu --opt flux_stamps=$it 2>&1 >/dev/null | grep -E "cycles|gather"
done
