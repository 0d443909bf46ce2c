#!/bin/bash
# Dev tool: bench lines under the pipeline's knobs (stream pairs x detector level streams), alternating on the same box.
run() { python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-side "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['stage_ms_alone'])"; }
for rep in 1 2; do
for wl in C3 C5; do
run --workload $wl --pipes 2 --det-sides 1
run --workload $wl --pipes 1 --det-sides 1
run --workload $wl --pipes 1 --det-sides 2
run --workload $wl --pipes 2 --det-sides 2
done
done
