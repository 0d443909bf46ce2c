# Dev tool: bench.py C2 by pipeline shape (pipes x depth x detector level streams), same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pipes
for cfg in "--pipes 2 --depth 3" "--pipes 3 --depth 4" "--pipes 1 --depth 2" "--pipes 2 --depth 4" "--pipes 3 --depth 3 --det-sides 2" "--pipes 2 --depth 3 --det-sides 2" "--pipes 4 --depth 5" "--pipes 2 --depth 3"; do
  echo "== $cfg"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-side --steps 40 $cfg > gpurun_out/pipes/o.json 2> gpurun_out/pipes/o.err || { tail -3 gpurun_out/pipes/o.err; exit 1; }
  python - <<PY
import json
d = json.loads(open("gpurun_out/pipes/o.json").readline())
print(d["value"], d["ms_per_step"], d["stage_ms_alone"], d["roofline"]["frac"], d.get("p50_batch_latency_ms"))
PY
done
