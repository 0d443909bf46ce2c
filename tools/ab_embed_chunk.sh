#!/bin/bash
# Dev tool: C2 bench lines by the embed forward's chunk size (256 faces = one workgroup per CU in the stage kernels; 128 leaves half the CUs
# to the detector while a stage kernel runs), alternating on one box.
run() { python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-side "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['stage_ms_alone'])"; }
for rep in 1 2; do
run
run --embed-chunk 128
run --embed-chunk 192
run --embed-chunk 128 --pipes 2
done
