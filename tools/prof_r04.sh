# Dev tool: the round-4 profiles (run on the GPU box from the repository root; results under gpurun_out/prof4, the summaries
# that are judged are copied into profiles/ by hand, named r04_*).
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof4
mkdir -p $O
cd $R
# 1. single-stream kernel stats of the bench: durations add up to the step
timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side --one-stream > $O/bench_single_stream.json 2> $O/bench_single_stream.err
# 2. HBM traffic passes over the embed net (separate --pmc passes)
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/fetch -- python3 tools/bench_embed.py 256 f16 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/write -- python3 tools/bench_embed.py 256 f16 > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write $O/r04_pmc_traffic.json > $O/pmc_traffic.txt 2>&1
# 3. the detector batch alone on one stream, launch by launch
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d $O/dettrace -- python3 tools/det_trace_run.py > $O/dettrace.log 2>&1
python3 tools/det_trace_sum.py $O/dettrace > $O/r04_detector_trace.txt 2>&1
# 4. SQ counters of the detector's new kernels (two passes over the same detector batches)
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/sqa -- python3 tools/det_trace_run.py > $O/sqa.log 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_VMEM -d $O/sqb -- python3 tools/det_trace_run.py > $O/sqb.log 2>&1
for k in ro_conv2_split_kernel ro_gemm_split_kernel crop_conv1_kernel pnet23_split_f16 pnet_conv1_kernel; do
  echo "== $k" >> $O/r04_detector_pmc_counters.txt
  python3 tools/pmc_sum.py $O/sqa $k >> $O/r04_detector_pmc_counters.txt 2>&1
  python3 tools/pmc_sum.py $O/sqb $k >> $O/r04_detector_pmc_counters.txt 2>&1
done
find $O -name "*kernel_stats.csv" | head
find $O -name "*.csv" -size +2000k -delete
find $O -name "*.db" -delete
du -sh $O
head -8 $O/pmc_traffic.txt; grep -A12 "^wall" $O/r04_detector_trace.txt
