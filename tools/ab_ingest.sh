#!/bin/bash
# Dev tool: the PCIe-inclusive C2 rate (every step's 398 MB uploaded from a pinned ring) by how the upload is issued
# (frame groups x copy streams, steps ahead), against the resident-frames rate of the same box.
for cfg in "resident 1 1 0" "pinned 1 1 0" "pinned 1 1 1" "pinned 1 1 2" "pinned 4 1 1" "pinned 1 1 0" "resident 1 1 0"; do
  set -- $cfg
  python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-side --ingest $1 --ingest-chunks $2 --ingest-streams $3 --ingest-ahead $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"
done
