"""Dev tool: do CU-partitioned HIP streams make the detector and the embedder overlap better than
the free-for-all?  hipExtStreamCreateWithCUMask through ctypes -> torch.cuda.ExternalStream; the C ABI takes a raw
hipStream_t, so nothing in libfrhip.so changes.  Measures each stage ALONE on a CU subset, then both TOGETHER (the
bench's two-stream step) on complementary subsets against the unmasked streams.

    python tools/bench_cumask.py
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
from facerecognition_infrenceengine_amd import FaceAnalysis, _lib
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_frames

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
NCU = 256


def masked_stream(cus):
    """stream restricted to the CU indices in ``cus``"""
    words = (ctypes.c_uint32 * (NCU // 32))()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), NCU // 32, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device="cuda:0")


def subset(n_of_8):
    """n_of_8 of every 8 consecutive CU indices (consecutive indices are dealt over the 8 XCDs: keeps every XCD in both halves)"""
    return [c for c in range(NCU) if (c // 8) % 8 < n_of_8] if False else [c for c in range(NCU) if (c % 8) < n_of_8]


torch.cuda.set_device(0)
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4)
    app.prepare(ctx_id=0)
batches = [synth_frames(64, 1080, 1920, i, dev) for i in range(2)]
crops = (torch.rand((256, 112, 112, 8), device=dev) * 2 - 1).half(); crops[..., 3:] = 0


def timed(fn, stream, n=6):
    with torch.cuda.stream(stream):
        for _ in range(2): fn()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# does the mask bite at all?  a plain torch GEMM on masked streams
ga = torch.randn((8192, 8192), device=dev, dtype=torch.float16); gb = torch.randn((8192, 8192), device=dev, dtype=torch.float16)
for n8 in (8, 4, 2):
    print(f"mask check: torch.matmul 8192^3 f16 on {n8 * 32:3d} CUs: {timed(lambda: torch.matmul(ga, gb), masked_stream(subset(n8))):.3f} ms", flush=True)
del ga, gb
app.det._nsides = 0
print("stage alone on a CU subset (ms per 64 x 1080p frames / per 256 faces):", flush=True)
for n8 in (8, 6, 5, 4, 3, 2):
    s = masked_stream(subset(n8))
    app.rec.use_stage14 = n8 == 8                      # the one-image-per-CU stage kernel needs all 256 CUs for 256 faces
    app.det.one_stream = True          # every level on the masked stream
    d = timed(lambda: app.det.detect_batch(batches[0], level_streams=1), s)
    app.det.one_stream = False
    e = timed(lambda: app.rec.forward(crops), s)
    print(f"  {n8 * 32:3d} CUs: detect {d:6.2f}  embed {e:6.2f}" + ("  (stage kernel)" if n8 == 8 else "  (layer by layer)"), flush=True)


def step_loop(s_det, s_emb, n=12):
    """the bench's schedule: detector of step i+1 on its stream beside the embed of step i"""
    def one(i):
        with torch.cuda.stream(s_emb):
            app.detect_embed_slots(batches[i % 2], det_stream=s_det)
    for i in range(3): one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): one(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("both stages, ONE (detector, embedder) stream pair (ms per step):", flush=True)
app.rec.use_stage14 = True
print(f"  unmasked streams, stage kernel:            {step_loop(torch.cuda.Stream(), torch.cuda.Stream()):6.2f}", flush=True)
app.rec.use_stage14 = False
print(f"  unmasked streams, layer by layer:          {step_loop(torch.cuda.Stream(), torch.cuda.Stream()):6.2f}", flush=True)
for nd in (2, 3, 4):
    s_det, s_emb = masked_stream(subset(nd)), masked_stream([c for c in range(NCU) if (c % 8) >= nd])
    print(f"  detector {nd * 32:3d} CUs | embedder {(8 - nd) * 32:3d} CUs (layer by layer): {step_loop(s_det, s_emb):6.2f}", flush=True)
