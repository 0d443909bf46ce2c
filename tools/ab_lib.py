"""Dev tool: same-box, same-process A/B of ONE C-ABI entry between the tree's libfrhip.so and a base build of it kept as
tools/ab/libfrhip_base.so (`git stash; make; cp libfrhip.so tools/ab/libfrhip_base.so; git stash pop; make`): the two are loaded side
by side through ctypes and called alternately on the same buffers, HIP events around every call, rounds interleaved.
usage: python tools/ab_lib.py stage28 | stage14 | stage14_f8 [faces]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from facerecognition_infrenceengine_amd import weights, _lib
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
what = sys.argv[1] if len(sys.argv) > 1 else "stage28"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
new = _lib.load()
base = C.CDLL(os.path.join(ROOT, "tools", "ab", "libfrhip_base.so"))
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
P, I = C.c_void_p, C.c_int
g = torch.Generator(device="cuda").manual_seed(3)
if what == "stage28":
    st = net.stage28
    x0 = (torch.randn((B, 28, 28, 128), generator=g, device="cuda") * 0.5).half()
    sig = [P, P, P, P, I, I, P]
    def call(lib, x, mid):
        return lib.fr_conv_stage28_f16(P(x.data_ptr()), P(mid.data_ptr()), P(st["w"].data_ptr()), P(st["prm"].data_ptr()), B, st["n"], P(torch.cuda.current_stream().cuda_stream))
    flops = 2.0 * B * 784 * 128 * 1152 * 2 * st["n"]
    base.fr_conv_stage28_f16.argtypes = sig; base.fr_conv_stage28_f16.restype = I
elif what == "stage14":
    st = net.stage14
    x0 = (torch.randn((B, 14, 14, 256), generator=g, device="cuda") * 0.5).half()
    sig = [P, P, P, P, I, I, P]
    def call(lib, x, y):
        return lib.fr_conv_stage14_f16(P(x.data_ptr()), P(y.data_ptr()), P(st["w"].data_ptr()), P(st["prm"].data_ptr()), B, st["n"], P(torch.cuda.current_stream().cuda_stream))
    flops = 2.0 * B * 196 * 256 * 2304 * 2 * st["n"]
    base.fr_conv_stage14_f16.argtypes = sig; base.fr_conv_stage14_f16.restype = I
elif what in ("crop_r", "crop_o"):
    # fr_crop_conv1_split (conv on the f16 matrix cores) over 64 x 1080p frames with every slot holding a random square box
    import bench
    from facerecognition_infrenceengine_amd import FaceAnalysis
    import warnings
    warnings.simplefilter("ignore")
    net_id = 0 if what == "crop_r" else 1
    det = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0).det
    frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
    cap = 512 if net_id == 0 else 64
    side = torch.rand((64, cap), generator=g, device="cuda") * 180 + 20
    cx = torch.rand((64, cap), generator=g, device="cuda") * 1900
    cy = torch.rand((64, cap), generator=g, device="cuda") * 1060
    boxes = torch.stack([cx - side / 2, cy - side / 2, cx + side / 2, cy + side / 2], -1).contiguous()
    counts = torch.full((64,), cap, dtype=torch.int32, device="cuda")
    w1, b1, s1 = det._rc1 if net_id == 0 else det._oc1
    p1 = 11 if net_id == 0 else 23
    x0 = torch.zeros((64 * cap, p1 * p1, 128), dtype=torch.uint8, device="cuda")
    sig = [I, P, I, I, I, P, P, I, P, P, P, P, I, P]
    def call(lib, x, y):
        return lib.fr_crop_conv1_split(net_id, P(frames.data_ptr()), 64, 1080, 1920, P(boxes.data_ptr()), P(counts.data_ptr()), cap,
                                       P(w1.data_ptr()), P(b1.data_ptr()), P(s1.data_ptr()), P(y.data_ptr()), 1, P(torch.cuda.current_stream().cuda_stream))
    flops = 1.0
    base.fr_crop_conv1_split.argtypes = sig; base.fr_crop_conv1_split.restype = I
else:
    raise SystemExit("unknown entry")
outs = {}
acc = {"base": [], "new": []}
for rnd in range(6):
    for name, lib in (("base", base), ("new", new)):
        ts = []
        for _ in range(5):
            x, y = x0.clone(), torch.empty_like(x0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rc = call(lib, x, y); e1.record(); torch.cuda.synchronize()
            assert rc == 0
            ts.append(e0.elapsed_time(e1))
        acc[name].append(sorted(ts)[len(ts) // 2])
        outs[name] = (x if what == "stage28" else y).clone()
    print(f"round {rnd}: base {acc['base'][-1] * 1e3:8.1f} us   new {acc['new'][-1] * 1e3:8.1f} us", flush=True)
same = torch.equal(outs["base"], outs["new"])
mb, mn = min(acc["base"]), min(acc["new"])
print(f"{what} {B} faces: base min {mb * 1e3:.1f} us ({flops / mb / 1e9:.0f} TF)  new min {mn * 1e3:.1f} us ({flops / mn / 1e9:.0f} TF)  "
      f"new/base {mn / mb:.4f}   outputs bit-equal: {same}")
