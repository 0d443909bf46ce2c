"""Dev tool: 1-ms buckets of detector / embedder / other kernel time from a rocprofv3 kernel trace (steady state)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
# the timed loop = the densest stretch: take the last conv_halo kernel and go back 120 ms
last = max(e[1] for e in ev if "conv_halo" in e[2])
t0 = last - 150_000_000
ev = [e for e in ev if e[0] >= t0 and e[1] <= last]
def kind(n):
    if "dconv" in n or "pnet" in n or "sort_nms" in n or "crop_resize" in n or "stage_select" in n or "box_refine" in n: return "D"
    if "conv_" in n or "fc_reduce" in n or "warp_affine" in n: return "E"
    if "zzz" in n or "pnet" in n or "sort_nms" in n or "crop_resize" in n or "stage_select" in n or "box_refine" in n: return "D"
    if "gallery" in n or "l2norm" in n or "match_decide" in n: return "M"
    if "copyBuffer" in n or "nccl" in n.lower(): return "C"
    return "o"
B = 1_000_000
nb = (last - t0) // B + 1
acc = [dict(E=0, D=0, M=0, C=0, o=0) for _ in range(nb)]
qs = [set() for _ in range(nb)]
for s, e, n, q in ev:
    k = kind(n)
    b0, b1 = (s - t0) // B, (e - t0) // B
    for b in range(b0, b1 + 1):
        lo, hi = max(s, t0 + b * B), min(e, t0 + (b + 1) * B)
        if hi > lo:
            acc[b][k] += hi - lo; qs[b].add(q)
for b in range(60, min(nb, 110)):
    a = acc[b]
    print(f"{b:4d} ms  D {a['D']/1e6:5.2f}  E {a['E']/1e6:5.2f}  M {a['M']/1e3:6.1f}us  C {a['C']/1e3:6.1f}us  o {a['o']/1e3:6.1f}us  queues {sorted(qs[b])}")
