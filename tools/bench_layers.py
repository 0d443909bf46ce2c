"""Dev tool: per-launch time of every conv of one r100 forward (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facerecognition_infrenceengine_amd import _lib
if os.environ.get("FR_DEBUG_LIB"):      # A/B of debug-build switches (FR_HALO_RESPF, ...) in the real net
    _lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), "libfrhip_debug.so"))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
for _ in range(3):
    net.forward(x)
acc = None
for rep in range(5):
    net.profile = []
    net.forward(x); torch.cuda.synchronize()
    t = [e0.elapsed_time(e1) * 1e3 for _, _, e0, e1 in net.profile]
    acc = t if acc is None else [min(a, b) for a, b in zip(acc, t)]
    meta = [(v, f) for v, f, _, _ in net.profile]
net.profile = None
groups = {}
for i, ((v, f), us) in enumerate(zip(meta, acc)):
    key = (v.split("(")[0][:44], round(f / 1e9, 1))
    g = groups.setdefault(key, [0, 0.0]); g[0] += 1; g[1] += us
print(f"{'kernel':46s} {'GF':>7s} {'n':>3s} {'us/launch':>9s} {'TF':>6s} {'ms tot':>7s}")
for (v, gf), (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"{v:46s} {gf:7.1f} {n:3d} {us / n:9.1f} {gf / (us / n) * 1e3:6.0f} {us / 1e3:7.3f}")
print("total ms", sum(acc) / 1e3)
