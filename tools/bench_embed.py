"""Dev tool: time the IResNet forward (and per-layer kernels) on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
arch = sys.argv[2] if len(sys.argv) > 2 else "r100"
net = IResNetHIP(weights.synth_iresnet_state(arch), arch)
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half()
x[..., 3:] = 0
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    net.forward(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{arch} B={B}: {ms:.3f} ms/forward, {B/ms*1e3:.0f} faces/s, {net.flops_per_face*B/ms/1e9:.1f} TFLOP/s")
