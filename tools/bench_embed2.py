"""Dev tool: does splitting the embed batch over two HIP streams (two half-batch forwards in flight) overlap the
memory phases of one launch with the MFMA phases of the other?"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = 256
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def one(n=10):
    for _ in range(n): net.forward(x)
def two(n=10, parts=2):
    h = B // parts
    xs = [x[i * h:(i + 1) * h].contiguous() for i in range(parts)]
    streams = [sa, sb][:parts] if parts == 2 else [torch.cuda.Stream() for _ in range(parts)]
    for _ in range(n):
        for s, xi in zip(streams, xs):
            with torch.cuda.stream(s):
                net.forward(xi)
for name, fn in (("one stream, B=256", one), ("two streams, 2 x 128", two), ("one stream, B=256", one), ("two streams, 2 x 128", two)):
    fn(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); fn(10)
    torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1)/10:.3f} ms per 256 faces", flush=True)
