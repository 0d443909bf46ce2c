"""Dev tool: phase cycle shares of the P-Net conv kernels (diagnostic stamps)."""
import os, sys, warnings
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
frames = (torch.rand((64, 1080, 1920, 3), device="cuda") * 255).to(torch.uint8)
det._s = _lib.stream_ptr()
H, W, s = 1080, 1920, 0.6
import math
hs, ws = math.ceil(H * s), math.ceil(W * s)
x1, h1, w1 = det._dconv(None, det.p1, 64, hs, ws, frames=frames)
for name, layer, x, h, w, fr in (("P1", det.p1, None, hs, ws, frames), ("P2", det.p2, x1, h1, w1, None)):
    st.zero_(); y, ho, wo = det._dconv(x, layer, 64, h, w, frames=fr); torch.cuda.synchronize()
    d = st.reshape(-1, 4).double(); d = d[d.sum(1) > 0]
    print(name, "waves", len(d), "cycles/wave: prefetch-issue %.0f  K-loop %.0f  epilogue %.0f  sync+store %.0f" % tuple(d.mean(0).tolist()))
    if name == "P2": x2, h2, w2 = y, ho, wo
st.zero_(); y, ho, wo = det._dconv(x2, det.p3, 64, h2, w2); torch.cuda.synchronize()
d = st.reshape(-1, 4).double(); d = d[d.sum(1) > 0]
print("P3", "waves", len(d), "cycles/wave: prefetch-issue %.0f  K-loop %.0f  epilogue %.0f  sync+store %.0f" % tuple(d.mean(0).tolist()))
