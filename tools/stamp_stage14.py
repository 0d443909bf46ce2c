"""Dev tool: in-kernel s_memtime segment sums of the 14x14 stage kernel (diagnostic build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")      # 256 workgroups x 8 waves x 8 values
from facerecognition_infrenceengine_amd import _lib as _fr_lib
_fr_lib.use_library(os.path.join(os.path.dirname(_fr_lib.LIB_PATH), "libfrhip_debug.so"))
from facerecognition_infrenceengine_amd import weights, _lib
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = 256
if len(sys.argv) > 1:
    os.environ["FR_S14_STAMP_LEVEL"] = sys.argv[1]
ABLS = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
st.zero_()
# FR_DBG_STAMPS is read by EVERY stamped kernel of the diagnostic library on every call, and the other kernels index the
# buffer by their own (much larger) grids: set it around the stage kernel's call only
orig = net._run_stage14


def stamped(h, B_):
    assert B_ * 8 * 8 <= st.numel()
    os.environ["FR_DBG_STAMPS"] = hex(st.data_ptr())
    try:
        return orig(h, B_)
    finally:
        del os.environ["FR_DBG_STAMPS"]


net._run_stage14 = stamped
names = ["wait vmcnt/lgkm", "barrier", "step body (MFMA+reads+DMA)", "epilogues", "prologues"]
nconv = 2 * net.stage14["n"]
steps = nconv * 72
for abl in ABLS:
    os.environ["FR_S14_ABL"] = str(abl % 100)
    os.environ["FR_S14_DEPHASE"] = str(abl // 100)            # abl = 100 * dephase + ablation bits
    st.zero_()
    net.forward(x)
    torch.cuda.synchronize()
    d = st.reshape(-1, 8)[: B * 8].double()
    tot = d[:, 5]
    print(f"ABL {abl} (1 no MFMA, 2 no reads, 4 no weight DMA, 8 no barrier): kernel cycles per wave mean {tot.mean():.0f}; clock "
          f"{(d[:, 5] / d[:, 6].clamp_min(1) * 100).mean():.0f} MHz; per conv {tot.mean().item() / nconv:.0f} cycles = "
          f"{(d[:, 6].mean().item() / 100) / nconv:.1f} us; K loop {d[:, 7].mean().item() / nconv:.0f} ({d[:, 7].mean().item() / steps:.0f} per step; MFMA-bound 832), "
          f"epilogue {d[:, 3].mean().item() / nconv:.0f}, prologue {d[:, 4].mean().item() / nconv:.0f}", flush=True)
    if os.environ.get("FR_S14_STAMP_LEVEL", "2") == "1":
        print(f"      epilogue segments per conv: wait+barrier {d[:, 0].mean().item() / nconv:.0f}, tiles {d[:, 1].mean().item() / nconv:.0f}, "
              f"lgkm+barrier {d[:, 2].mean().item() / nconv:.0f}, rest (HBM copy of second convs) {(d[:, 3] - d[:, 0] - d[:, 1] - d[:, 2]).mean().item() / nconv:.0f}")
    if abl == 0 and os.environ.get("FR_S14_STAMP_LEVEL", "2") == "2":
        for k, nme in enumerate(names[:3]):
            print(f"   {nme:30s} {d[:, k].mean().item() / steps:9.1f} cycles per step")
        for w in range(8):
            dw = d[w::8]
            print(f"   wave {w}: wait {dw[:, 0].mean() / steps:6.1f} barrier {dw[:, 1].mean() / steps:6.1f} body {dw[:, 2].mean() / steps:6.1f}")
