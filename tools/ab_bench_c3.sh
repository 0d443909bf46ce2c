cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for cfg in "--det-sides 1 --det-level-nms per-level" "--det-sides 2 --det-level-nms merged" "--det-sides 1 --det-level-nms merged" "--det-sides 2 --det-level-nms per-level"; do
    echo "== C3 $cfg"
    timeout -k 10 200 python bench.py --workload C3 --steps 40 --warmup 4 --no-cpu-baseline --no-side $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d.get('stage_ms_alone'))"
  done
done
