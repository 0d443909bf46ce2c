"""Dev tool: phase cycle shares of the R-Net / O-Net conv kernels (diagnostic stamps, s_memtime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(4096 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["FR_DBG_STAMPS"] = hex(st.data_ptr())
# stamps exist only in the diagnostic twin of the library (make -C facerecognition_infrenceengine_amd/csrc debug)
from facerecognition_infrenceengine_amd import _lib as _fr_lib
_fr_lib.use_library(os.path.join(os.path.dirname(_fr_lib.LIB_PATH), "libfrhip_debug.so"))
from facerecognition_infrenceengine_amd import weights, _lib
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
det = MTCNNHIP(*weights.synth_mtcnn_states(), device="cuda:0")
det._s = _lib.stream_ptr()
def run(name, layer, x, B, h, w, items_per_block):
    for _ in range(2):
        y, ho, wo = det._dconv(x, layer, B, h, w)
    st.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y, ho, wo = det._dconv(x, layer, B, h, w); e1.record(); torch.cuda.synchronize()
    d = st.reshape(-1, 4).double(); d = d[d.sum(1) > 0]
    m = (d.mean(0) / items_per_block).tolist()
    print(f"{name}: {e0.elapsed_time(e1)*1e3:7.1f} us  per item cycles/wave: prefetch-issue {m[0]:.0f}  K-loop {m[1]:.0f}  epilogue(pool) {m[2]:.0f}  sync+store {m[3]:.0f}  total {sum(m):.0f}")
    return y, ho, wo
B2 = 64 * 512
x = torch.randn((B2, 24, 24, 4), device="cuda"); x[..., 3] = 0
y, h, w = run("R1", det.r1, x, B2, 24, 24, 4)
y, h, w = run("R2", det.r2, y, B2, h, w, 1)
B3 = 64 * 64
x = torch.randn((B3, 48, 48, 4), device="cuda"); x[..., 3] = 0
y, h, w = run("O1", det.o1, x, B3, 48, 48, 6)
y, h, w = run("O2", det.o2, y, B3, h, w, 1)
y, h, w = run("O3", det.o3, y, B3, h, w, 1)
