"""Dev tool: batch size from which the walk64 kernel beats the per-tile halo kernel on the Cin = 64 layers (r100 forward, ms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights, iresnet
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
iresnet.WALK64_SKIP = range(0)
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
for B in (64, 96, 128, 144, 160, 192, 224, 256):
    x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
    out = []
    for use in (True, False, True, False):
        net.use_walk64 = use
        for _ in range(3): net.forward(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): net.forward(x)
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / 10, 3))
    print(B, "walk64 / halo / walk64 / halo ms:", out, flush=True)
