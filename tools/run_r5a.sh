#!/bin/bash
# round 5, first GPU call: the GPU suite (with the new batch-path oracle tests), the two probes, a baseline bench line
set -o pipefail
O=gpurun_out/r5a; mkdir -p $O
step() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $O/progress.log
  timeout -k 10 $secs "$@" > $O/$name.log 2>&1; local rc=$?
  echo "rc=$rc" | tee -a $O/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/progress.log; exit $rc; fi
  return 0
}
step tests 900 python -m pytest tests -m gpu -x -q
step bench_c2 300 python bench.py --steps 20 --warmup 4
step probe 300 python tools/nms_order_probe.py
step crossover 300 python tools/crossover_det.py
tail -3 $O/tests.log; tail -c 600 $O/bench_c2.log
