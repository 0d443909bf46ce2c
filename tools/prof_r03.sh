set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof3
cd $R
# 1. single-stream kernel stats of the bench
timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/prof3/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side --one-stream > $R/gpurun_out/prof3/bench_single_stream.json 2> $R/gpurun_out/prof3/bench_single_stream.err
# 2. HBM traffic passes over the embed net
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof3/fetch -- python3 tools/bench_embed.py 256 f16 > $R/gpurun_out/prof3/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof3/write -- python3 tools/bench_embed.py 256 f16 > $R/gpurun_out/prof3/write.log 2>&1
# 3. SQ counters of the stage kernel, two passes
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/prof3/sqa -- python3 tools/prof_stage14.py 4 > $R/gpurun_out/prof3/sqa.log 2>&1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $R/gpurun_out/prof3/sqb -- python3 tools/prof_stage14.py 4 > $R/gpurun_out/prof3/sqb.log 2>&1
python3 tools/pmc_traffic.py $R/gpurun_out/prof3/fetch $R/gpurun_out/prof3/write $R/gpurun_out/prof3/r03_pmc_traffic.json > $R/gpurun_out/prof3/pmc_traffic.txt 2>&1
python3 tools/pmc_sum.py $R/gpurun_out/prof3/sqa conv_stage14 > $R/gpurun_out/prof3/sq_counters.txt 2>&1
python3 tools/pmc_sum.py $R/gpurun_out/prof3/sqb conv_stage14 >> $R/gpurun_out/prof3/sq_counters.txt 2>&1
python3 tools/pmc_sum.py $R/gpurun_out/prof3/sqa conv_stage28 > $R/gpurun_out/prof3/sq_counters_stage28.txt 2>&1
python3 tools/pmc_sum.py $R/gpurun_out/prof3/sqb conv_stage28 >> $R/gpurun_out/prof3/sq_counters_stage28.txt 2>&1
find $R/gpurun_out/prof3 -name "*kernel_stats.csv" | head
# keep only small files
find $R/gpurun_out/prof3 -name "*.csv" -size +2000k -delete
find $R/gpurun_out/prof3 -name "*.db" -delete
du -sh $R/gpurun_out/prof3
cat $R/gpurun_out/prof3/sq_counters.txt; head -12 $R/gpurun_out/prof3/pmc_traffic.txt
