"""Dev tool: r100 forward time (ms, HIP events, the prepared fr_conv_sequence path) for 1..8 faces with the 3x3 / s1 convs in the
in-block split-K form (conv_inblock.hip) and in the split-K + epilogue form - where is the crossover (IResNetHIP.inblock_batch)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
net = IResNetHIP(weights.synth_iresnet_state("r100", seed=1234), "r100", "cuda:0")
g = torch.Generator().manual_seed(0)
for B in (1, 2, 4, 5, 6, 8, 12, 16):
    x = torch.zeros((B, 112, 112, 8), dtype=torch.float16, device="cuda")
    x[..., :3] = (torch.rand((B, 112, 112, 3), generator=g) * 2 - 1).half().cuda()
    out = []
    for ib in (16, 0, 16, 0):
        net.inblock_batch = ib
        net.release_plans()
        for _ in range(5):
            net.forward(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            net.forward(x)
        e1.record(); torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / 30, 3))
    print(B, "faces: in-block / split-K / in-block / split-K ms:", out, flush=True)
