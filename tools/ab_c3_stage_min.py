"""Dev tool: bench.py --workload C3 (8 x 4K frames, 128 faces per step) with the stage kernels' batch thresholds as they are (144: the
128-face embed runs layer by layer) and lowered to 128 (one workgroup per face on HALF the CUs: slower alone, but the other half is
free for the detector of the next step)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    from facerecognition_infrenceengine_amd import iresnet
    iresnet.STAGE14_MIN_BATCH = int(sys.argv[1]); iresnet.STAGE28_MIN_BATCH = int(sys.argv[2])
    sys.argv = ["bench.py", "--workload", "C3", "--no-cpu-baseline", "--no-side"]
    import bench
    bench.main()
    sys.exit(0)
for cfg in (("144", "144"), ("128", "128"), ("128", "144"), ("144", "144"), ("128", "128")):
    r = subprocess.run([sys.executable, __file__, *cfg], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("stage14 / stage28 from", cfg, ":", d["value"], d["ms_per_step"], d["stage_ms_alone"], flush=True)
    except Exception as e:
        print(cfg, "failed", r.stderr[-400:], flush=True)
