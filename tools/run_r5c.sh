#!/bin/bash
set -o pipefail
O=gpurun_out/r5c; mkdir -p $O
step() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $O/progress.log
  timeout -k 10 $secs "$@" > $O/$name.log 2>&1; local rc=$?
  echo "rc=$rc" | tee -a $O/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/progress.log; exit $rc; fi
  return 0
}
step front 400 python tools/ab_front_chunk.py 256
step probe 300 python tools/nms_order_probe.py
step bench_c3 300 python bench.py --workload C3 --steps 20 --warmup 4 --cpu-frames 1
step bench_c3_g1 300 python bench.py --workload C3 --steps 20 --warmup 4 --no-cpu-baseline --embed-group 1
step bench_c1 400 python bench.py --workload C1 --steps 40 --warmup 4 --cpu-threads 1,16,0
cat $O/front.log | tail -12
