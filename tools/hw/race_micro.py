"""Dev tool: run the packed-f32 victim beside each aggressor kernel family (see race_micro.hip)."""
import ctypes as C, os, sys, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "librace_micro.so"))
P = C.c_void_p
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(1024, device="cuda")
names = ["mfma16x16x32_f16", "mfma32x32x16_f16", "mfma16x16x16_f16", "mfma16x16x4_f32", "valu_pk_f16", "ds_read_b128", "none"]
for mode, mname in ((0, "pk_mul+pk_add"), (1, "pk_fma"), (2, "scalar control")):
    for kind, kname in enumerate(names):
        errs = torch.zeros(64, dtype=torch.int32, device="cuda"); tot = torch.zeros(2, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            if kname != "none":
                for _ in range(6):
                    lib.race_aggressor(kind, 200000, 2048, P(sink.data_ptr()), P(sb.cuda_stream))
        with torch.cuda.stream(sa):
            for _ in range(40):
                lib.race_victim(mode, 2000, 1024, P(errs.data_ptr()), P(tot.data_ptr()), P(sa.cuda_stream))
        torch.cuda.synchronize()
        e = errs.cpu().tolist()
        print(f"victim {mname:14s} vs {kname:18s}: bad waves/lane max {max(e):6d}  lanes hit {[i for i, v in enumerate(e) if v][:6]}..{[i for i, v in enumerate(e) if v][-3:]}  lo/hi errs {tot.cpu().tolist()}")
