# Dev tool: bench.py C2 of the tree under _old/ (an older commit, built) and of this tree, alternating, same box
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for t in _old .; do echo "== $t"; (cd $t && timeout -k 10 200 python bench.py --no-cpu-baseline --no-side 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stage_ms_alone'], d['roofline']['frac'])") || exit 1; done
done
