# Dev tool: the round's bench lines (C2 default, C5 and C3 WITH the CPU oracle leg, C4 on one GPU), one after the other on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench4
timeout -k 10 500 python bench.py > gpurun_out/bench4/c2.json 2> gpurun_out/bench4/c2.err && \
timeout -k 10 500 python bench.py --workload C5 --cpu-frames 4 --cpu-threads 16 > gpurun_out/bench4/c5.json 2> gpurun_out/bench4/c5.err && \
timeout -k 10 500 python bench.py --workload C3 --cpu-frames 1 --cpu-threads 16 > gpurun_out/bench4/c3.json 2> gpurun_out/bench4/c3.err && \
timeout -k 10 500 python bench.py --workload C4 --no-cpu-baseline --no-side > gpurun_out/bench4/c4.json 2> gpurun_out/bench4/c4.err
for f in c2 c5 c3 c4; do python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bench4/$f.json").readline())
    print("$f", d["value"], d["ms_per_step"], d["stage_ms_alone"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("oracle_check", {}).get("ok"), d.get("latency_c1_ms"), d.get("value_pcie"))
except Exception as e:
    print("$f", "failed", e)
PY
done
