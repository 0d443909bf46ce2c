"""Dev tool: the LAST replayed single-frame get() of a rocprofv3 kernel trace of tools/trace_c1.py as a timeline: start (us from the
call's first kernel), duration, queue, kernel - to see what the call's critical path is made of."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
big = sorted(i for i in range(len(rows) - 1) if st[i + 1] - en[i] > 300000)
seg = rows[big[-2] + 1:big[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
qs = {}
for r in seg:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    n = re.sub(r"\(.*", "", n)[:56]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{q:<2d} {n}")
print("kernels", len(seg), "span us", (max(int(r["End_Timestamp"]) for r in seg) - t0) / 1e3)
