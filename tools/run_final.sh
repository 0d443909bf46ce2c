#!/bin/bash
# final check of the tree on a GPU box: smoke, the GPU suite, the default bench line
set -o pipefail
O=gpurun_out/r5final; mkdir -p $O
step() {
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $O/progress.log
  timeout -k 10 $secs "$@" > $O/$name.log 2>&1; local rc=$?
  echo "rc=$rc" | tee -a $O/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/progress.log; exit $rc; fi
  return 0
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step tests 1000 python -m pytest tests -m gpu -x -q
step bench 400 python bench.py
tail -2 $O/smoke.log; tail -3 $O/tests.log; tail -c 400 $O/bench.log
