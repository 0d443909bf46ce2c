cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dt
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d $O/dettrace -- python3 tools/det_trace_run.py > $O/dettrace.log 2>&1
python3 tools/det_trace_sum.py $O/dettrace > $O/detector_trace.txt 2>&1
find $O -name "*.csv" -size +2000k -delete
find $O -name "*.db" -delete
grep -A30 "^wall" $O/detector_trace.txt
