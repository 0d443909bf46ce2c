"""Dev tool: the 14x14 stage as one launch (fr_conv_stage14_f16) against the layer-by-layer path, r100, B faces.
Interleaved rounds in one process (rule: perf deltas come from interleaved rounds in ONE process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights, _lib
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0


def timed(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


res = {"stage": [], "layer": []}
for r in range(4):
    for mode in ("stage", "layer"):
        net.use_stage14 = mode == "stage"
        res[mode].append(timed(lambda: net.forward(x)))
net.use_stage14 = True
for mode, v in res.items():
    print(f"r100 forward, {B} faces, 14x14 stage as {mode:5s}: " + " ".join(f"{t:.3f}" for t in v) + " ms", flush=True)
# the stage kernel alone on a random 14x14x256 map
st = net.stage14
h = (torch.randn((B, 14, 14, 256), device="cuda") * 0.5).half()
y = torch.empty_like(h)
lib = net.lib
t = timed(lambda: lib.fr_conv_stage14_f16(_lib.ptr(h), _lib.ptr(y), _lib.ptr(st["w"]), _lib.ptr(st["prm"]), B, st["n"], _lib.stream_ptr()), 10)
fl = 2.0 * B * 196 * 256 * 2304 * 2 * st["n"]
print(f"stage kernel alone: {t:.3f} ms for {2 * st['n']} convs = {t / (2 * st['n']) * 1e3:.1f} us per conv, {fl / t / 1e9:.0f} TFLOP/s "
      f"= {fl / t / 1e9 / 2500:.3f} of 2.5 PF", flush=True)
net.profile = []
net.forward(x); torch.cuda.synchronize()
per = {}
for v, f_, a, b in net.profile:
    d = per.setdefault(v, [0, 0.0, 0.0]); d[0] += 1; d[1] += f_; d[2] += a.elapsed_time(b)
net.profile = None
for v, (n, f_, tt) in sorted(per.items(), key=lambda kv: -kv[1][2]):
    print(f"   {v:62s} x{n:3d} {tt:7.3f} ms  {f_/tt/1e9:7.0f} TFLOP/s")
