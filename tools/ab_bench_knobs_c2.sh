#!/bin/bash
# Dev tool: C2 bench lines under the pipeline's knobs (stream pairs x detector level streams x steps in flight), alternating on the same box.
run() { python bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-side "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['stage_ms_alone'])"; }
for rep in 1 2; do
run --pipes 1 --det-sides 2
run --pipes 1 --det-sides 1
run --pipes 1 --det-sides 3
run --pipes 2 --det-sides 2
run --pipes 1 --det-sides 2 --depth 4
done
