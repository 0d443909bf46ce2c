# Dev tool -> profiles/r05_soak.txt: long pipelined runs; bench.py's self_check compares EVERY timed step's ids / decisions / counts with a sequential
# single-stream re-run of the batch that step processed, planted_top1 counts the faces that came back with their planted row
O=gpurun_out/soak; mkdir -p $O
for cfg in "C2 --steps 1500" "C5 --steps 800" "C3 --steps 1500" "C2 --steps 600 --ingest pinned" "C4 --steps 600 --force-exchange"; do
  set -- $cfg; w=$1; shift
  name=$(echo "$cfg" | tr ' -' '__')
  timeout -k 10 500 python bench.py --workload $w "$@" --warmup 4 --no-cpu-baseline --no-side > $O/$name.json 2> $O/$name.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT at $cfg"; exit $rc; fi
  python - <<PY
import json
try:
    d = json.loads(open("$O/$name.json").readline())
    print("bench.py --workload $cfg: rc $rc, %s faces/s, %s ms/step; self_check: %s; planted_top1 %d of %d" % (d["value"], d["ms_per_step"], d["self_check"], d["planted_top1"]["matched_own_row"], d["planted_top1"]["faces"]))
except Exception as e:
    print("bench.py --workload $cfg: rc $rc, no line:", e)
PY
done
