"""Dev tool: a few r100 forwards at 256 faces for rocprofv3 (--kernel-trace --stats, or --pmc passes) of the stage kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((256, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    net.forward(x)
torch.cuda.synchronize()
