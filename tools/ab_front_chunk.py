"""Dev tool: the front of the embed net (stem .. 56 -> 28 entry) in groups of faces that fit the memory-side cache
(IResNetHIP.front112_chunk / front56_chunk): r100 forward ms at B faces, variants interleaved over rounds in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
variants = [(0, 0), (64, 128), (64, 256), (128, 128), (64, 64), (128, 256), (0, 128), (96, 192), (52, 104)]
ref = None
acc = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        net.front112_chunk, net.front56_chunk = v
        for _ in range(2): net.forward(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): e = net.forward(x)[0]
        e1.record(); torch.cuda.synchronize()
        acc[v].append(e0.elapsed_time(e1) / 6)
        if ref is None: ref = e.clone()
        same = torch.equal(e, ref)             # grouping changes no arithmetic unless a group falls into another batch-size mode
        cos = float((1 - torch.nn.functional.cosine_similarity(e, ref)).max())
        print(f"round {rnd} front112 {v[0]:4d} front56 {v[1]:4d}: {acc[v][-1]:.3f} ms  bit-equal {same}  max 1-cos {cos:.1e}", flush=True)
for v in variants:
    print(f"front112 {v[0]:4d} front56 {v[1]:4d}: " + " ".join(f"{t:.3f}" for t in acc[v]) + f"   min {min(acc[v]):.3f} ms", flush=True)
