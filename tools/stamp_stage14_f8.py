"""Dev tool: in-kernel s_memtime segment sums of the fp8 14x14 stage kernel (diagnostic build)."""
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
net.enable_fp8(x[:32].contiguous())
for _ in range(3):
    net.forward(x)
torch.cuda.synchronize()
orig = net._run_stage14_f8


def stamped(h, h8, B_):
    os.environ["FR_DBG_STAMPS"] = hex(st.data_ptr())       # around this call only: every stamped kernel of the library reads it
    try:
        return orig(h, h8, B_)
    finally:
        del os.environ["FR_DBG_STAMPS"]


net._run_stage14_f8 = stamped
nconv = 2 * net.stage14["n"]
for rep in range(2):
    st.zero_()
    net.forward(x)
    torch.cuda.synchronize()
    d = st.reshape(-1, 8)[: B * 8].double()
    tot = d[:, 5]
    print(f"kernel cycles per wave mean {tot.mean():.0f}; clock {(d[:, 5] / d[:, 6].clamp_min(1) * 100).mean():.0f} MHz; longest wave "
          f"{d[:, 6].max().item() / 100:.1f} us; per conv {tot.mean().item() / nconv:.0f} cycles: K loop {d[:, 7].mean().item() / nconv:.0f} "
          f"({d[:, 7].mean().item() / nconv / 18:.0f} per step; MFMA-bound 1664), epilogue {d[:, 3].mean().item() / nconv:.0f}, "
          f"prologue (residual / zero + first fragments) {d[:, 4].mean().item() / nconv:.0f}", flush=True)
