"""Dev tool: in-kernel s_memtime segment sums of the halo conv (diagnostic build, shares only)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["FR_DBG_STAMPS"] = hex(st.data_ptr())
from tools import bench_conv
B = 256
for shape in ((14, 256, 256), (28, 128, 128)):
    st.zero_()
    bench_conv.run(B, shape[0], shape[0], shape[1], shape[2], iters=3, tag="stamped")
    torch.cuda.synchronize()
    nb = B if shape[0] == 14 else B * 4
    d = st.reshape(-1, 8)[: nb * 8, :5].double()
    tot = d.sum(1)
    names = ["h0 mfma+reads", "waits(lgkm,vm)", "barrier", "issue W/X", "h1 mfma+reads"]
    print(shape, "per-step cycles (s_memtime ticks, 100MHz?) mean per wave:")
    steps = 9 * shape[1] // 64
    for k, nme in enumerate(names):
        print(f"   {nme:16s} {d[:, k].mean().item() / steps:10.1f}  share {d[:, k].sum().item() / tot.sum().item():.3f}")
    print("   total/step", tot.mean().item() / steps)
