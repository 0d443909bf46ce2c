"""Dev tool: from a rocprofv3 kernel trace of bench.py - over the last 150 ms of the timed loop, how much of the wall time has NO
kernel running, exactly one stream's kernels, or kernels of several queues at once; and the largest idle gaps."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
last = max(e[1] for e in ev if "conv_stage14" in e[2])
t0 = last - 150_000_000
ev = [e for e in ev if e[0] >= t0 and e[1] <= last]
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = {0: 0, 1: 0, 2: 0}
cur, prev = 0, t0
gaps = []
for t, dlt in pts:
    k = min(cur, 2)
    busy[k] += t - prev
    if cur == 0 and t - prev > 20_000:
        gaps.append((t - prev, prev - t0))
    cur += dlt; prev = t
tot = sum(busy.values())
print(f"window {tot/1e6:.1f} ms: no kernel {busy[0]/tot*100:.1f} %, one kernel {busy[1]/tot*100:.1f} %, two or more {busy[2]/tot*100:.1f} %")
gaps.sort(reverse=True)
print("largest idle gaps (us, at ms):", [(round(g / 1e3, 1), round(a / 1e6, 2)) for g, a in gaps[:12]])
print("idle gaps > 20 us:", len(gaps), "sum", round(sum(g for g, _ in gaps) / 1e6, 2), "ms")
