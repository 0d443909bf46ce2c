# Dev tool: SQ counters of the stage kernels over a 256-face r100 forward (two --pmc passes of 8 counters) -> gpurun_out/prof5sq
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof5sq; mkdir -p $O; cd $R
run() { local secs=$1; shift; timeout -k 10 $secs "$@"; local rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; exit $rc; fi; return 0; }
run 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O/sqa -- python3 tools/prof_stage14.py 4 > $O/sqa.log 2>&1
run 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/sqb -- python3 tools/prof_stage14.py 4 > $O/sqb.log 2>&1
for k in conv_stage14 conv_stage28 conv_walk64 "conv_halo_kernel<2, 13, 384" "conv_mfma_kernel<2"; do
  echo "== $k" >> $O/sq_counters.txt
  python3 tools/pmc_sum.py $O/sqa "$k" >> $O/sq_counters.txt 2>&1
  python3 tools/pmc_sum.py $O/sqb "$k" >> $O/sq_counters.txt 2>&1
done
find $O -name "*.csv" -size +3000k -delete; find $O -name "*.db" -delete
cat $O/sq_counters.txt
