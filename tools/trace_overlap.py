"""Dev tool: from a rocprofv3 kernel trace, wall time vs union-busy time vs summed kernel time, and the largest idle gaps."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")) for r in rows)
# keep the last 60 % (steady state)
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 0.4
ev = [e for e in ev if e[0] >= t_lo]
wall = ev[-1][1] - ev[0][0]
tot = sum(e[1] - e[0] for e in ev)
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"wall {wall/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms ({100*busy/wall:.1f} %)  summed kernel time {tot/1e6:.2f} ms  (concurrency {tot/busy:.2f}x)")
print("queues used:", sorted({e[3] for e in ev}))
gaps.sort(reverse=True)
print("largest idle gaps (us, next kernel):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:6]], " total idle", round(sum(g for g, _ in gaps) / 1e6, 2), "ms")
names = {}
for s, e, n, q in ev:
    if "nccl" in n.lower() or "rccl" in n.lower():
        names.setdefault(n, []).append((e - s) / 1e3)
for n, v in names.items():
    print("collective kernel", n, "n", len(v), "avg us", round(sum(v) / len(v), 1), "max", round(max(v), 1))
