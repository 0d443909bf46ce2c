# Dev tool: SQ counters of one detector kernel (name pattern $1) over tools/det_trace_run.py, two passes; prints the LARGEST
# dispatch's counters (pyramid level 0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcdet; mkdir -p $O
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/a -- python3 tools/det_trace_run.py > $O/a.log 2>&1 &&
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_VMEM -d $O/b -- python3 tools/det_trace_run.py > $O/b.log 2>&1
python3 - "$1" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
for d in ("gpurun_out/pmcdet/a", "gpurun_out/pmcdet/b"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = sorted(v)
        print(f"{k:28s} n={len(v):3d} max={v[-1]:.4e} (the 4 largest: {[f'{x:.3e}' for x in v[-4:]]})")
PY
find $O -name "*.csv" -size +2000k -delete; find $O -name "*.db" -delete
