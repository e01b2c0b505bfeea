#!/bin/bash
# usage (GPU box): bash tools/step_ab.sh  -- cfg4 bf16 training step (graph replay), same box, interleaved variants:
#   default | fused hidden layers for the 256-wide nets only | none (two launches per layer) | zero-filled unused gradients
run() { echo -n "$1: "; env $2 timeout 400 python bench.py --mode train --precision bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms graph,', round(d.get('eager_ms_per_step') or 0,3), 'ms eager, loss', round(d['loss'],4))"; }
for i in 1 2; do
  run "default                 " "MODA_X=0"
  run "MODA_BWD256=256         " "MODA_BWD256=256"
  run "MODA_BWD256=0           " "MODA_BWD256=0"
  run "MODA_MATERIALIZE_GRADS=1" "MODA_MATERIALIZE_GRADS=1"
done
