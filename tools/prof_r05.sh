# Dev tool: the round-5 records (run on the GPU box from the repository root; results under gpurun_out/prof5, the summaries that are
# judged are copied into profiles/ by hand, named r05_*).  Steps are joined so that a timeout stops the script.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof5
mkdir -p $O
cd $R
run() { local secs=$1; shift; timeout -k 10 $secs "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return 0; }
# 0. the bench lines, one after the other on this box
run 500 python bench.py > $O/c2.json 2> $O/c2.err
run 500 python bench.py --workload C5 --cpu-frames 4 --cpu-threads 16 > $O/c5.json 2> $O/c5.err
run 500 python bench.py --workload C3 --cpu-frames 1 --cpu-threads 16 > $O/c3.json 2> $O/c3.err
run 500 python bench.py --workload C4 --no-cpu-baseline --no-side > $O/c4.json 2> $O/c4.err
# 1. single-stream kernel stats of the bench: durations add up to the step
run 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side --one-stream > $O/bench_single_stream.json 2> $O/bench_single_stream.err
# 1b. the DEFAULT (pipelined) command under the profiler
run 500 rocprofv3 --output-format csv --kernel-trace --stats -d $O/statsd -- python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
# 2. HBM traffic passes over the embed net (separate --pmc passes)
run 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 tools/bench_embed.py 256 f16 > $O/fetch.log 2>&1
run 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 tools/bench_embed.py 256 f16 > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write $O/r05_pmc_traffic.json > $O/pmc_traffic.txt 2>&1
# 3. the detector batch alone on one stream, launch by launch
run 300 rocprofv3 --output-format csv --kernel-trace -d $O/dettrace -- python3 tools/det_trace_run.py > $O/dettrace.log 2>&1
python3 tools/det_trace_sum.py $O/dettrace > $O/r05_detector_trace.txt 2>&1
# 4. the embed net layer by layer (HIP events)
run 300 python3 tools/bench_embed.py 256 f16 detail > $O/r05_embed_layers.txt 2>&1
run 300 python3 tools/bench_embed.py 256 fp8 >> $O/r05_embed_layers.txt 2>&1
find $O -name "*kernel_trace.csv" -size +2000k -delete
find $O -name "*.csv" -size +4000k -delete
find $O -name "*.db" -delete
du -sh $O
for f in c2 c5 c3 c4; do tail -c 300 $O/$f.json; echo; done
head -8 $O/pmc_traffic.txt
