"""Dev tool: in-kernel s_memtime segment sums of the 28x28 stage kernel (diagnostic build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")      # 256 workgroups x 8 waves x 8 values
from facerecognition_infrenceengine_amd import _lib as _fr_lib
_fr_lib.use_library(os.path.join(os.path.dirname(_fr_lib.LIB_PATH), "libfrhip_debug.so"))
from facerecognition_infrenceengine_amd import weights, _lib
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = 256
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
orig = net._run_stage28


def stamped(h, B_, nb):
    assert B_ * 8 * 8 <= st.numel()
    os.environ["FR_DBG_STAMPS"] = hex(st.data_ptr())       # around this call only: every stamped kernel of the library reads it
    try:
        return orig(h, B_, nb)
    finally:
        del os.environ["FR_DBG_STAMPS"]


net._run_stage28 = stamped
npass = 4 * net.stage28["n"]
names = ["load wait (pass start)", "prologue reads", "K loop (36 steps)", "post-loop barrier + epilogue", "drain + next halo issue (per conv)"]
for rep in range(2):
    st.zero_()
    net.forward(x)
    torch.cuda.synchronize()
    d = st.reshape(-1, 8)[: B * 8].double()
    tot = d[:, 5]
    print(f"longest wave {d[:, 6].max().item() / 100:.1f} us; kernel cycles per wave mean {tot.mean():.0f}; clock "
          f"{(d[:, 5] / d[:, 6].clamp_min(1) * 100).mean():.0f} MHz; per pass {tot.mean().item() / npass:.0f} cycles = "
          f"{(d[:, 6].mean().item() / 100) / npass:.2f} us")
    for k, nme in enumerate(names):
        print(f"   {nme:40s} {d[:, k].mean().item() / npass:9.0f} cycles per pass" + (f"  ({d[:, k].mean().item() / npass / 36:.0f} per step; MFMA-bound 800)" if k == 2 else ""), flush=True)
