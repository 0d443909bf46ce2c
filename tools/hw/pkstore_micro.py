"""Dev tool: v_pk_mul_f32 -> global_store victim beside (a) nothing, (b) a small MFMA loop, (c) a
large-register MFMA loop shaped like the lean conv kernel, (d) the product's conv kernels.  ONE pass, no retries."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from facerecognition_infrenceengine_amd import _lib
lib = _lib.load()
mic = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpkstore_micro.so"))
P = C.c_void_p
ITERS, BLOCKS, REP = 64, 1024, 200
out = torch.empty((ITERS, BLOCKS * 256, 2), device="cuda")
hist = torch.zeros(64, dtype=torch.int32, device="cuda"); total = torch.zeros(2, dtype=torch.int32, device="cuda")
sink = torch.zeros(1 << 20, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
# conv aggressor operands (stage-3 body conv, 256 faces)
B = 256
x = torch.randn((B, 14, 14, 256), device="cuda").half(); w = (torch.randn((256, 2304), device="cuda") * 0.02).half()
y = torch.empty((B, 14, 14, 256), dtype=torch.float16, device="cuda"); bias = torch.zeros(256, device="cuda")
ca = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(bias), None, None, None, B, 14, 14, 256, 256, 3, 3, 1, 1, 14, 14, 0, 1)
x7 = torch.randn((B, 7, 7, 512), device="cuda").half(); w7 = (torch.randn((512, 4608), device="cuda") * 0.02).half()
y7 = torch.empty((B, 7, 7, 512), dtype=torch.float16, device="cuda"); b7 = torch.zeros(512, device="cuda")
c7 = _lib.ConvArgs(_lib.ptr(x7), _lib.ptr(w7), _lib.ptr(y7), _lib.ptr(b7), None, None, None, B, 7, 7, 512, 512, 3, 3, 1, 1, 7, 7, 0, 1)

def aggress(kind):
    with torch.cuda.stream(sb):
        if kind == "mfma_small": mic.pk_aggr(0, 400000, 2048, P(sink.data_ptr()), P(sb.cuda_stream))
        elif kind == "mfma_big": mic.pk_aggr(1, 40000, 1024, P(sink.data_ptr()), P(sb.cuda_stream))
        elif kind == "valu_big_no_mfma": mic.pk_aggr(2, 100000, 1024, P(sink.data_ptr()), P(sb.cuda_stream))
        elif kind == "mfma_small_512thr": mic.pk_aggr(3, 400000, 1024, P(sink.data_ptr()), P(sb.cuda_stream))
        elif kind == "mfma_big_256thr": mic.pk_aggr(4, 40000, 2048, P(sink.data_ptr()), P(sb.cuda_stream))
        elif kind == "f32_mfma_512thr": mic.pk_aggr(5, 200000, 1024, P(sink.data_ptr()), P(sb.cuda_stream))
        elif kind == "conv_halo":
            for _ in range(400): lib.fr_conv_nhwc_f16(C.byref(ca), P(sb.cuda_stream))
        elif kind == "conv_mfma":
            for _ in range(400): lib.fr_conv_nhwc_f16(C.byref(c7), P(sb.cuda_stream))

names = {0: "pk_mul->store", 1: "pk_mul,s_nop7->store", 2: "pk_mul(sgpr,op_sel_hi)->store", 3: "v_mul x2->store (control)", 4: "pk_mul->pk_add->store", 5: "exec restore; pk_mul->store",
         6: "pk_add IN PLACE, src1 halves swapped", 7: "pk_add swapped, dst separate", 8: "pk_add in place, not swapped", 9: "pk_add in place swapped, pk_mov, store",
         10: "pk_MUL, src1 halves swapped", 11: "pk_add, SRC0 halves swapped", 12: "pk_MOV, lo <- src0.hi, hi <- src1.lo",
         13: "pk_FMA, src1 halves swapped", 14: "pk_add op_sel_hi:[1,0] only (hi <- src1.lo)", 15: "pk_add op_sel:[0,1] only (lo <- src1.hi)"}
MODES = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
for kind in (("none", "mfma_big", "valu_big_no_mfma", "mfma_small_512thr", "mfma_big_256thr", "f32_mfma_512thr") if os.environ.get("PK_AGGR") else ("none", "mfma_big", "conv_halo") if os.environ.get("PK_SHORT") else ("none", "mfma_small", "mfma_big", "conv_halo", "conv_mfma")):
    for mode in MODES:
        torch.cuda.synchronize(); hist.zero_(); total.zero_()
        aggress(kind)
        with torch.cuda.stream(sa):
            for r in range(REP):
                mic.pk_victim(mode, ITERS, BLOCKS, P(out.data_ptr()), P(sa.cuda_stream))
                mic.pk_check(mode, ITERS, BLOCKS, P(out.data_ptr()), P(hist.data_ptr()), P(total.data_ptr()), P(sa.cuda_stream))
        torch.cuda.synchronize()
        h = hist.cpu().tolist()
        q = [sum(h[i * 16:(i + 1) * 16]) for i in range(4)]
        print(f"aggressor {kind:18s} victim {names[mode]:44s}: mismatches {int(total[0]):9d}  by lane quarter {q}  "
              f"words that hold the UNSWAPPED sum {int(total[1]):9d}", flush=True)
