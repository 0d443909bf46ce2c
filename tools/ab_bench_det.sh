#!/bin/bash
# Dev tool: end-to-end A/B of the detector's level-stream / level-NMS options inside bench.py (alternating repeats).
cd /root/repo
for rep in 1 2; do
  for cfg in "--det-sides 1 --det-level-nms per-level" "--det-sides 1 --det-level-nms merged" "--det-sides 2 --det-level-nms merged" "--det-sides 2 --det-level-nms per-level"; do
    echo "== $cfg"
    timeout -k 10 200 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-side $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d.get('stage_ms_alone'), d['roofline']['frac'], d.get('oracle_check'))"
  done
done
