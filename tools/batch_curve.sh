# Dev tool -> profiles/r05_batch_size_curve.txt: faces/s, ms/step and batch latency of the C2 pipeline by frames per step (one box, back to back)
O=gpurun_out/curve; mkdir -p $O
for n in 4 8 16 32 64 128; do
  timeout -k 10 300 python bench.py --frames $n --steps 30 --warmup 4 --no-cpu-baseline --no-side > $O/f$n.json 2> $O/f$n.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT at $n"; exit $rc; fi
  python - <<PY
import json
try:
    d = json.loads(open("$O/f$n.json").readline())
    print("frames/step %4d: %8.1f faces/s  %7.3f ms/step  p50 batch latency %7.2f ms (3 in flight)  detect %.2f  align+embed %.2f ms alone  batch path %s" % ($n, d["value"], d["ms_per_step"], d["p50_batch_latency_ms"], d["stage_ms_alone"]["detect"], d["stage_ms_alone"]["align_embed"], d["detector_path"]["batch"]))
except Exception as e:
    print("frames/step $n failed", e)
PY
done
for n in 4 8 16 32 64 128; do
  timeout -k 10 300 python bench.py --frames $n --steps 30 --warmup 4 --no-cpu-baseline --no-side --depth 1 > $O/d1_f$n.json 2> $O/d1_f$n.err
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT at $n"; exit $rc; fi
  python - <<PY
import json
try:
    d = json.loads(open("$O/d1_f$n.json").readline())
    print("depth 1, frames/step %4d: %8.1f faces/s  %7.3f ms/step  p50 batch latency %7.2f ms (1 in flight)" % ($n, d["value"], d["ms_per_step"], d["p50_batch_latency_ms"]))
except Exception as e:
    print("depth 1 frames/step $n failed", e)
PY
done
