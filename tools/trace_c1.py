"""Dev tool: kernel timeline of ONE single-frame get() (640x480, cap_o = 1) under HIP-graph replay.
Run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o c1 -- python3 tools/trace_c1.py`, then
`python3 tools/trace_c1.py --report DIR/c1_kernel_trace.csv`."""
import os, sys, time, warnings, csv, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the replays are the trailing repetitions: take the last N kernels where N = kernels per replay, found from the marker gap
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    gaps = [(st[i + 1] - en[i], i) for i in range(len(rows) - 1)]
    big = sorted(i for g, i in gaps if g > 300000)          # > 0.3 ms: host time between calls
    lo, hi = big[-2] + 1, big[-1] + 1
    seg = rows[lo:hi]
    s0, e1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    busy, cur_e = 0, s0
    for r in seg:                                            # union of kernel intervals
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if b > cur_e:
            busy += b - max(a, cur_e); cur_e = b
    print(f"kernels {len(seg)}  span {(e1 - s0) / 1e3:.1f} us  union busy {busy / 1e3:.1f} us  sum {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3:.1f} us")
    per = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = r["Kernel_Name"][:70]
        per[k][0] += 1; per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{t / 1e3:9.1f} us {n:4d}  {k}")
    sys.exit(0)
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x", cap_o=1).prepare(ctx_id=0)
frame = synth_frame(480, 640, 7)
app.enable_graphs(True)
ts = []
for i in range(12):
    t0 = time.perf_counter()
    faces = app.get(frame)
    ts.append((time.perf_counter() - t0) * 1e3)
    time.sleep(0.002)
print("faces", len(faces), "get() ms p50", round(float(np.percentile(ts[4:], 50)), 3))
