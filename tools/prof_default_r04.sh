# Dev tool: rocprofv3 kernel stats of the DEFAULT bench command (pipelined streams), beside the single-stream profile of prof_r04.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof4d
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete
ls -la $O/stats/*/
tail -c 600 $O/bench_default.json
