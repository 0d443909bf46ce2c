# Dev tool: the round's bench lines (C2 default, C5, C4 on one GPU, C3), one after the other on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench3
timeout -k 10 400 python bench.py > gpurun_out/bench3/c2.json 2> gpurun_out/bench3/c2.err && \
timeout -k 10 400 python bench.py --workload C5 --no-cpu-baseline > gpurun_out/bench3/c5.json 2> gpurun_out/bench3/c5.err && \
timeout -k 10 400 python bench.py --workload C4 --no-cpu-baseline --no-side > gpurun_out/bench3/c4.json 2> gpurun_out/bench3/c4.err && \
timeout -k 10 400 python bench.py --workload C3 --no-cpu-baseline --no-side > gpurun_out/bench3/c3.json 2> gpurun_out/bench3/c3.err
for f in c2 c5 c4 c3; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench3/$f.json").readline())
    print("$f", d["value"], d["ms_per_step"], d["stage_ms_alone"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("oracle_check", {}).get("ok"), d.get("latency_c1_ms"), d.get("value_pcie"))
except Exception as e:
    print("$f", "failed", e)
PY
done
