"""Dev tool: the LAST detector batch of a rocprofv3 kernel trace of tools/det_trace_run.py, kernel by kernel in launch order."""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a batch ends with the O-Net stage's NMS, the second sort_nms<512, 512> of the batch: the last batch starts behind the
# third such launch from the end
idx = [i for i, r in enumerate(rows) if "sort_nms<512, 512>" in r["Kernel_Name"]]
rows = rows[(idx[-3] + 1 if len(idx) >= 3 else 0):idx[-1] + 1]
while rows and "pnet_conv1" not in rows[0]["Kernel_Name"]:
    rows.pop(0)
t0 = int(rows[0]["Start_Timestamp"])
tot = {}
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] = tot.get(n, 0) + d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:8.1f} us  grid {r.get('Grid_Size', '?'):>9}  {n}")
print("wall %.1f us, kernel sum %.1f us" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, sum(tot.values())))
for n, d in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"{d:9.1f} us  {n}")
