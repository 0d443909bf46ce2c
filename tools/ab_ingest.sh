#!/bin/bash
# Dev tool: the PCIe-inclusive C2 rate (every step's 398 MB uploaded from a pinned ring) by how the upload is issued
# (steps ahead) and by the stream pairs, against the resident-frames rate of the same box.
run() { python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-side "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run --ingest resident
run --ingest pinned --ingest-ahead 2
run --ingest pinned --ingest-ahead 3
run --ingest pinned --ingest-ahead 2 --pipes 2
run --ingest pinned --ingest-ahead 2 --depth 4
run --ingest pinned --ingest-ahead 3 --depth 4
done
