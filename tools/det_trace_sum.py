"""Dev tool: the LAST detector batch of a rocprofv3 kernel trace of tools/det_trace_run.py, kernel by kernel in launch order."""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pnet_conv1_kernel" in r["Kernel_Name"]]
# batches start at a pnet_conv1 launch that follows a non-P-Net kernel stretch: take the last run of 12 levels
starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 40]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
tot = {}
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] = tot.get(n, 0) + d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:8.1f} us  grid {r.get('Grid_Size', '?'):>9}  {n}")
print("wall %.1f us, kernel sum %.1f us" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, sum(tot.values())))
for n, d in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"{d:9.1f} us  {n}")
