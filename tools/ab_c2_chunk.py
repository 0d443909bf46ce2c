"""Dev tool: bench.py C2 with the embed net's chunk at 256 faces (one stage-kernel launch over every CU) and at 128 (two launches on half
the CUs each: the same CU-time, the other half free for the detector of the next step)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from facerecognition_infrenceengine_amd import iresnet
    orig = iresnet.IResNetHIP.__init__
    chunk = int(sys.argv[1])
    def init(self, *a, **k):
        orig(self, *a, **k)
        self.max_chunk = chunk
    iresnet.IResNetHIP.__init__ = init
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-side"]
    import bench
    bench.main()
    sys.exit(0)
for c in ("256", "128", "256", "128", "192"):
    r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("max_chunk", c, ":", d["value"], d["ms_per_step"], d["stage_ms_alone"], flush=True)
    except Exception as e:
        print(c, "failed", r.stderr[-400:], flush=True)
