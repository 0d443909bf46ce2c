"""Dev tool: two half-batch forwards on two streams, short run for a kernel trace (do the launches overlap?)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((256, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
xa, xb = x[:128].contiguous(), x[128:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    with torch.cuda.stream(sa): net.forward(xa)
    with torch.cuda.stream(sb): net.forward(xb)
torch.cuda.synchronize()
