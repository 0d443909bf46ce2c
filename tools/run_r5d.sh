#!/bin/bash
set -o pipefail
O=gpurun_out/r5d; mkdir -p $O
step() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $O/progress.log
  timeout -k 10 $secs "$@" > $O/$name.log 2>&1; local rc=$?
  echo "rc=$rc" | tee -a $O/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/progress.log; exit $rc; fi
  return 0
}
step ab28 300 python tools/ab_lib.py stage28 256
step embed_tests 600 python -m pytest tests/test_gpu_embed.py -x -q -m gpu
tail -3 $O/ab28.log; tail -3 $O/embed_tests.log
