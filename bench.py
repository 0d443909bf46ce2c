#!/usr/bin/env python3
"""Headline benchmark: faces/sec end-to-end at 1080p (BASELINE.json metric, config C2).

One step = one batch of 64 synthetic 1080p BGR frames, resident in HBM, through the whole hot
path on one GPU: MTCNN pyramid + NMS -> 5-point warp -> ArcFace IResNet-100 (f16 MFMA) ->
gallery match against 10 000 rows, ids on the host at the end of the step.  With N ranks each
GPU processes its own 64-frame batch (weak scaling); the gallery is row-sharded and the
embedding rows are all-gathered over RCCL before the shared match.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (the IResNet-100 14x14 stage on
MFMA: 58 convs in one launch), timed with HIP events on its launch stream in an instrumented pass after the timed region;
`cpu_baseline` is the CPU oracle (a port, not the reference) timed on a bounded sample - the first frames of the GPU's
own batch 0 - at 1, 16 and all host threads (the best is reported); the same run is the `oracle_check` of the GPU's
boxes / embeddings / ids for those frames, and a mismatch makes the run exit non-zero.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The step runs on 2 x (detector, embedder, pyramid side) streams + a copy stream.  The HIP runtime maps streams onto
# 4 hardware queues by default; streams that share a queue serialise (measured: 2 pipes on 4 queues = 20.2 ms/step,
# on 8 queues = 17.5 ms; with RCCL's own streams beside them 8 queues alias again: 19.0 ms, 12 or 16 queues: 17.9 ms).
# Final build of round 2, one box, N=1: 8 queues 13.32 ms/step, 12: 13.21, 16: 13.21, 24: 13.07; three alternating repeats on
# another box: 16 queues 13.16 / 13.01 / 13.19, 24 queues 13.04 / 13.00 / 12.87, 32 queues 12.99 -> 24.
# Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise); the launcher's
# environment normally carries it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FRAMES, H, W = 64, 1080, 1920
GALLERY_ROWS = 10_000
FACES_PER_FRAME = 4                    # O-Net cap (C2 keeps F = 4 -> 256 faces/batch)
MFMA_PEAK_TFLOPS = 2500.0              # dense f16/bf16 (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS_F8 = 5000.0           # dense fp8 (block-scaled MFMA with unit scales)
PMC_FILE = "profiles/r05_pmc_traffic.json"


def synth_frames(n, h, w, seed, device):
    """Structured synthetic BGR frames (low-pass + mid + fine noise), uint8, generated on device."""
    g = torch.Generator(device=device).manual_seed(seed)
    import torch.nn.functional as F
    low = torch.rand((n, 3, h // 16, w // 16), generator=g, device=device)
    mid = torch.rand((n, 3, h // 4, w // 4), generator=g, device=device)
    img = 0.6 * F.interpolate(low, (h, w), mode="bicubic", align_corners=False) \
        + 0.3 * F.interpolate(mid, (h, w), mode="bilinear", align_corners=False) \
        + 0.1 * torch.rand((n, 3, h, w), generator=g, device=device)
    return (img.permute(0, 2, 3, 1).clamp(0, 1) * 255).round().to(torch.uint8).contiguous()


def cpu_baseline(frames, gallery, threads=None, match_memo=None, warmups=0, reps=1):
    """CPU oracle on a bounded sample of the same workload (rank 0, N == 1 only): ``frames`` (list of uint8 HWC BGR
    arrays - copies of the first frames of the GPU's batch 0) through oracle detect -> align -> r100 fp32 -> the
    literal per-row Python match loop over ``gallery`` (ordered dict str(row) -> f32[512]: the rows the GPU scans, keyed
    by strings as the reference's person ids are).  ``threads``: torch intra-op threads (None = the process default =
    all host cores).  ``match_memo`` (a dict shared by the calls of a thread sweep): the match loop is single-threaded
    Python whatever ``threads`` is, so it runs - and is timed - in the FIRST call only; later calls add that time to their
    own detect + embed time and take its ids (a 1.25 M-row gallery is millions of interpreter iterations per face).
    ``warmups`` / ``reps``: untimed passes, then the MEDIAN of ``reps`` timed passes (SURVEY.md 8(d) for C1: 3 and 20; the C2
    sample is ~10 - 30 s of CPU work per pass and is timed once).
    Returns (record, per-frame results)."""
    from facerecognition_infrenceengine_amd import weights
    from oracle import align as oalign, detect as odetect, match as omatch, nets as onets
    default_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    p, r, o = weights.synth_mtcnn_states()
    st = weights.synth_iresnet_state("r100")
    memo = match_memo if match_memo is not None else {}
    frame_memo = memo.setdefault("frames", {})        # frame index -> (ids, decisions, seconds the literal loop took for them)

    def one_pass(use_memo):
        t0 = time.perf_counter()
        faces, results, t_match, t_reused = 0, [], 0.0, 0.0
        for fi, fr in enumerate(frames):
            b, s, k = odetect.detect(fr, p, r, o, cap_o=FACES_PER_FRAME)
            rec = {"bbox": b, "score": s, "kps": k, "emb": np.zeros((0, 512), np.float32), "ids": [], "dec": []}
            if len(s):
                crops = [oalign.norm_crop(fr, kk)[0] for kk in k]
                x = torch.from_numpy(np.stack([oalign.crop_to_net(c) for c in crops]))
                rec["emb"] = onets.iresnet_forward(st, x, weights.IRESNET_LAYERS["r100"]).numpy()
                hit = frame_memo.get(fi) if use_memo else None
                if hit is not None and len(hit[0]) == len(s):
                    rec["ids"], rec["dec"] = hit[0], hit[1]
                    t_reused += hit[2]              # only the frames whose loop really was skipped add their recorded time
                else:
                    tm = time.perf_counter()
                    for e in rec["emb"]:
                        q = omatch.renormalise(e / np.linalg.norm(e))
                        bid, bs = omatch.linear_scan(q, gallery)                 # literal per-row Python loop
                        rec["ids"].append(-1 if bid is None else int(bid))
                        rec["dec"].append(omatch.decide_live(bid, bs)[0] is not None)
                    tf = time.perf_counter() - tm
                    t_match += tf
                    if use_memo:
                        frame_memo[fi] = (rec["ids"], rec["dec"], tf)
                faces += len(s)
            results.append(rec)
        return time.perf_counter() - t0 + t_reused, faces, results, t_match + t_reused, t_reused > 0

    for _ in range(warmups):
        one_pass(False)
    passes = [one_pass(reps == 1) for _ in range(max(reps, 1))]
    passes.sort(key=lambda t: t[0])
    dt, faces, results, t_match, reused = passes[len(passes) // 2]
    torch.set_num_threads(default_threads)
    how = ", partly as timed in an earlier run of the sweep" if reused else ""
    rep_note = f"median of {reps} passes after {warmups} warm-up passes, " if reps > 1 else ""
    return {"value": round(faces / dt, 3), "unit": "faces/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
            "ms_per_pass": round(dt * 1e3, 2),
            "sample": f"{len(frames)} synthetic {H}x{W} frames (the first frames of the GPU's batch 0), {faces} faces, r100 "
                      f"fp32 torch-CPU + literal {len(gallery)}-row Python match loop "
                      f"({t_match:.3f} s, single-threaded{how}), {rep_note}{dt:.2f} s"}, results


def oracle_check(results, gpu, cos_tol):
    """The CPU oracle's per-frame results against what the GPU produced for the same frames of batch 0
    (/root/reference/infrenceServer.py:528-552: faces, embeddings, best id, decision)."""
    out = {"frames": len(results), "faces": 0, "counts_equal": True, "boxes_max_err": 0.0, "kps_max_err": 0.0,
           "score_max_err": 0.0, "min_cos": 1.0, "ids_equal": True, "decisions_equal": True}
    cap = gpu["bbox"].shape[1]
    for f, rec in enumerate(results):
        n = len(rec["score"])
        if n != int(gpu["counts"][f]):
            out["counts_equal"] = False
            continue
        out["faces"] += n
        for j in range(n):
            out["boxes_max_err"] = max(out["boxes_max_err"], float(np.abs(rec["bbox"][j] - gpu["bbox"][f, j]).max()))
            out["kps_max_err"] = max(out["kps_max_err"], float(np.abs(rec["kps"][j] - gpu["kps"][f, j]).max()))
            out["score_max_err"] = max(out["score_max_err"], float(abs(rec["score"][j] - gpu["det_score"][f, j])))
            a, b = rec["emb"][j].astype(np.float64), gpu["embedding"][f * cap + j].astype(np.float64)
            out["min_cos"] = min(out["min_cos"], float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))))
            out["ids_equal"] &= rec["ids"][j] == int(gpu["idx"][f * cap + j])
            out["decisions_equal"] &= bool(rec["dec"][j]) == (int(gpu["dec"][f * cap + j]) == 1)
    out["ok"] = bool(out["counts_equal"] and out["ids_equal"] and out["decisions_equal"] and out["faces"] > 0
                     and out["boxes_max_err"] <= 5e-3 and out["kps_max_err"] <= 5e-3 and out["score_max_err"] <= 5e-5
                     and out["min_cos"] >= 1.0 - cos_tol)
    out["tolerances"] = {"boxes_px": 5e-3, "kps_px": 5e-3, "det_score": 5e-5, "one_minus_cos": cos_tol, "ids": "exact"}
    for k in ("boxes_max_err", "kps_max_err", "score_max_err"):
        out[k] = float(f"{out[k]:.3g}")
    out["min_cos"] = round(out["min_cos"], 7)
    return out


def c1_latency(app, device, n=30):
    """BASELINE config C1 on the GPU: ONE 640x480 frame (host uint8), one face slot, 100-row gallery, through the
    reference-shaped calls `get(frame)` + match; p50 of n calls, ms.  Eager and with HIP-graph replay."""
    from facerecognition_infrenceengine_amd import GalleryMatcher
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import synth_frame
    frame = synth_frame(480, 640, 7)
    app1 = app.clone_with(cap_o=1)
    g = torch.Generator(device=device).manual_seed(2)
    gm1 = GalleryMatcher(device)
    gm1.set_rows(range(100), torch.randn((100, 512), generator=g, device=device), normalise=True)
    out = {}
    for tag, graphs in (("get_match_p50", False), ("get_match_graph_p50", True)):
        app1.enable_graphs(graphs)
        ts = []
        for i in range(n + 5):
            t0 = time.perf_counter()
            faces = app1.get(frame)
            if faces:
                gm1.match(np.stack([f.normed_embedding for f in faces]))
            ts.append((time.perf_counter() - t0) * 1e3)
        out[tag] = round(float(np.percentile(ts[5:], 50)), 3)
        out["faces"] = len(faces)
    app1.enable_graphs(False)
    return out


class stdout_to_stderr:
    """RCCL prints a version banner on file descriptor 1 when a communicator is created; the driver wants ONE JSON line
    on stdout.  Route fd 1 to fd 2 around the process-group set-up (and its first collective)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run as a
    CHILD process - this process has not touched the GPU (importing torch does not) - pass rank 0's JSON line
    through on stdout and exit with the launcher's code (non-zero if any rank failed)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def ranks_sharing_a_device(ranks_rec):
    """[(rank, rank)] pairs of a gathered ``ranks`` record that report the same device (PCI address, else uuid, else index)"""
    seen, clash = {}, []
    for d in ranks_rec:
        key = d.get("pci_bus_id") or d.get("uuid") or ("index", d["device_index"])
        if key in seen:
            clash.append((seen[key], d["rank"]))
        seen.setdefault(key, d["rank"])
    return clash


def rank_identity(device):
    """What proves that a rank sat on a device of its own: the HIP device index it used, the PCI address and the uuid of that
    device (torch's device properties; a field this torch build does not carry is left out)."""
    pr = torch.cuda.get_device_properties(device)
    out = {"device_index": device.index, "name": pr.name}
    if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        out["pci_bus_id"] = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    if hasattr(pr, "uuid"):
        out["uuid"] = str(pr.uuid)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU oracle leg (timed at each thread count)")
    ap.add_argument("--cpu-threads", default="1,16,0", help="torch thread counts of the CPU leg (0 = all host cores); the "
                                                             "best is reported as cpu_baseline, all of them in cpu_baseline_sweep")
    ap.add_argument("--cpu-reps", type=int, default=0, help="timed passes of the CPU leg per thread count, the median is reported "
                                                            "(default: 20 after 3 warm-up passes for C1 - SURVEY.md 8(d) - and 1 otherwise)")
    ap.add_argument("--no-side", action="store_true", help="skip the side measurements (value_pcie, latency_c1_ms)")
    ap.add_argument("--depth", type=int, default=3, help="steps in flight before the oldest one's ids are fetched")
    ap.add_argument("--ingest", default="resident", choices=["resident", "pinned"],
                    help="resident: frames already in HBM (the headline metric); pinned: every step's frames cross "
                         "PCIe from a pinned host ring on a copy stream (reported in DESIGN.md, never the headline)")
    ap.add_argument("--ingest-ahead", type=int, default=2,
                    help="pinned ingest: a step's upload is issued this many steps before the step itself is enqueued (the capture side hands a "
                         "batch over when its ring slot is full, not when the GPU asks for it).  Measured, same box (tools/ab_ingest.sh): resident "
                         "22 940 faces/s; pinned, ahead 0 / 1 / 2: 20 300 / 21 360 / 21 560; 4 frame groups on 1 / 2 copy streams: no gain / -5 %%")
    ap.add_argument("--frames", type=int, default=0,
                    help="side measurement: frames per step instead of the workload's own (C2: 64) - the serving curve of latency against "
                         "throughput by batch size (profiles/r05_batch_size_curve.txt); config.workload states the number used")
    ap.add_argument("--gallery-rows", type=int, default=0, help="override the gallery size (C4: 1 000 000 rows in total)")
    ap.add_argument("--gallery", default="f32", choices=["f32", "f16", "f8"],
                    help="f16 / f8: one-pass coarse scan of a 16- / 8-bit copy on the f16 / fp8 matrix cores + exact f32 "
                         "re-rank of the top 4 / 8")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with one rank: still run the two all-gathers of the sharded match over RCCL (rehearsal)")
    ap.add_argument("--pipes", type=int, default=0,
                    help="independent (detector, embedder) stream pairs, used round-robin.  Default: 1 for the 64-frame batches, 2 for "
                         "C3's 8 x 4K.  Measured with round 4's detector (tools/ab_bench_knobs.sh, same box, pipes x detector level "
                         "streams): C2 2x1 23 400 faces/s, 1x1 24 150, 1x2 25 400, 2x2 24 000; C5 25 500 / 25 470 / 27 070 / 25 060; "
                         "C3 20 780 / 18 800 / 18 340 / 20 880 - with one pair the one-workgroup-per-CU stage kernels of step i find the "
                         "CUs free of step i + 2's detector blocks")
    ap.add_argument("--embed-group", type=int, default=0,
                    help="A/B: consecutive steps whose face slots share ONE embed forward (FaceAnalysis.detect_embed_slots(crops_out=...) + "
                         "embed_slots).  Default 1.  Measured for C3 (a step = 8 x 4K frames x 16 slots = 128 faces = the stage kernels on half "
                         "of the CUs), same box: 1: 20 620 faces/s (5.37 ms/step), 2: 19 030 (5.82) - the CUs a 128-workgroup stage launch "
                         "leaves free are where the NEXT step's detector runs meanwhile (detect 2.7 + align/embed 4.3 alone = 7.0 ms against "
                         "5.4 per pipelined step); a 256-face forward owns every CU and the step becomes the sum (profiles/r05_c3_embed_group.txt)")
    ap.add_argument("--one-stream", action="store_true", help="detector and embedder on one stream (no overlap)")
    ap.add_argument("--det-sides", type=int, default=None, help="A/B: side streams the detector deals pyramid levels 1.. over")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--workload", default=None, choices=["C1", "C2", "C3", "C4", "C5"],
                    help="BASELINE.json config.  Default: C2 on one GPU (the headline: 64x1080p, 4 faces/frame, 10k rows) "
                         "and C4 on several (the same frames per GPU, ONE shared 1 000 000-row gallery row-sharded over "
                         "the ranks, f16 one-pass scan + exact re-rank, embedding rows all-gathered over RCCL); C1 (1 x "
                         "640x480, 1 face, 100 rows), C3 (8 x 4K, 16 faces/frame) and C5 (fp8 embed convs + fp8 scan of "
                         "a 10 M-row gallery row-sharded 8 ways: 1.25 M rows per GPU) are side measurements for DESIGN.md")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "C4" if int(os.environ.get("WORLD_SIZE", str(max(args.gpus, 1)))) > 1 else "C2"
    global FRAMES, H, W, FACES_PER_FRAME, GALLERY_ROWS
    if args.workload == "C1":
        FRAMES, H, W, FACES_PER_FRAME, GALLERY_ROWS = 1, 480, 640, 1, 100
    elif args.workload == "C3":
        FRAMES, H, W, FACES_PER_FRAME = 8, 2160, 3840, 16
    if args.frames > 0:
        FRAMES = args.frames
    world_env = int(os.environ.get("WORLD_SIZE", str(max(args.gpus, 1))))
    if args.workload == "C5":           # 10 M rows over 8 GPUs; fewer ranks keep the per-GPU shard (1.25 M rows)
        GALLERY_ROWS = 1_250_000 * world_env
        args.gallery = "f8"
    if args.workload == "C4":           # BASELINE config 4: ONE shared 1 M-row gallery, whatever the number of ranks
        GALLERY_ROWS = 1_000_000
        if args.gallery == "f32":
            args.gallery = "f16"
    if args.gallery_rows:
        GALLERY_ROWS = args.gallery_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    if world == 1 and (args.force_exchange or os.environ.get("FR_INIT_PG") == "1"):           # rehearsal: the N > 1 code path (RCCL collectives) with one rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        with stdout_to_stderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            dist.barrier()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)
            dist.barrier()

    import warnings
    from facerecognition_infrenceengine_amd import FaceAnalysis, GalleryMatcher, _lib
    from facerecognition_infrenceengine_amd.distributed import HipOps, ShardedGalleryMatcher, shard_rows
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        app = FaceAnalysis(name="synthetic", arch="r100", cap_o=FACES_PER_FRAME)
        app.prepare(ctx_id=local_rank)

    nbatch = min(max(args.steps, 1), 3)
    batches = [synth_frames(FRAMES, H, W, rank * 1000 + i, device) for i in range(nbatch)]
    if args.workload == "C5":               # fp8 body convs, calibrated on the faces of the first batch
        n8 = app.calibrate_fp8(batches[0][:16])
        assert n8 > 0

    # gallery: seeded unit rows, row-sharded over the ranks; every rank draws ONLY its own shard (seed = shard start)
    lo, hi = shard_rows(GALLERY_ROWS, world, rank)
    g = torch.Generator(device=device).manual_seed(1 + lo)
    Gs = torch.randn((hi - lo, 512), generator=g, device=device)
    # planted matches: every face of this rank's batches owns one row of this rank's shard =
    # its embedding + N(0, 0.02) noise (renormalised by set_rows): top-1 is well separated (score ~ 0.9 >> 0.4), so the
    # timed ids are meaningful, not near-tie argmaxes of random rows
    q_max = FRAMES * FACES_PER_FRAME
    gp = torch.Generator().manual_seed(77 + rank)
    n_plant = min(nbatch * q_max, hi - lo)
    plant_rows = torch.randperm(hi - lo, generator=gp)[:n_plant].to(device)          # distinct local rows
    planted = []                            # per batch: global row per slot (-1: empty slot / no room)
    for b in range(nbatch):
        r0 = app.detect_embed_slots(batches[b])
        valid = (torch.arange(FACES_PER_FRAME, device=device)[None, :] < r0["counts"][:, None]).reshape(-1)
        rows_b = torch.full((q_max,), -1, dtype=torch.int64, device=device)
        take = plant_rows[b * q_max:(b + 1) * q_max]
        sel = valid.nonzero().squeeze(1)[:take.numel()]
        rows_b[sel] = take[:sel.numel()]
        e = r0["normed_embedding"][sel]
        noise = torch.randn(e.shape, generator=g, device=device) * 0.02
        Gs[rows_b[sel]] = e + noise
        planted.append(torch.where(rows_b >= 0, rows_b + lo, rows_b).cpu())
    gm = GalleryMatcher(device, scan=args.gallery)
    gm.set_rows(range(lo, hi), Gs, normalise=True)
    del Gs
    sharded = ShardedGalleryMatcher(HipOps(gm, lo), q_max, force_exchange=args.force_exchange)

    ingest = None
    if args.ingest == "pinned":
        from facerecognition_infrenceengine_amd.ingest import FrameIngest
        ingest = FrameIngest(FRAMES, H, W, device, depth=args.depth + max(args.embed_group, 1) + args.ingest_ahead)
        for k in range(ingest.depth):       # what the capture side would have written
            ingest.host_buffer(k)[...] = batches[k % nbatch].cpu().numpy()

    # two HIP streams: the detector cascade of step i+1 (latency-bound) runs beside the embed convs of step i
    # (MFMA-bound); align/embed/match of a step wait for its own detector through an event
    two = not args.one_stream
    app.det.one_stream = args.one_stream        # profiling: pyramid levels on one stream too, per-kernel durations add up
    if args.det_sides is not None:
        app.det.level_streams = args.det_sides
    if args.pipes <= 0:
        args.pipes = 1 if FRAMES >= 16 else 2
    if args.embed_group <= 0:
        args.embed_group = 1
    G = args.embed_group
    pipes = []
    for _ in range(args.pipes if two else 1):
        pipes.append((torch.cuda.Stream(device=device) if two else None,
                      torch.cuda.Stream(device=device) if two else torch.cuda.current_stream(device)))

    # ids leave the device through pinned host buffers + an event: a pageable .cpu() drains both streams
    q_rows = FRAMES * FACES_PER_FRAME
    pinned = [{"idx": torch.empty(q_rows, dtype=torch.int64).pin_memory(),
               "dec": torch.empty(q_rows, dtype=torch.int32).pin_memory(),
               "counts": torch.empty(FRAMES, dtype=torch.int32).pin_memory()} for _ in range(args.depth + G)]
    # embed groups (G > 1): the aligned crops of G consecutive steps side by side, one buffer per stream pair (a pair's launches
    # are ordered: the next group's warps come behind this group's embed forward)
    group_crops = [torch.empty((G * q_rows, 112, 112, 8), dtype=torch.float16, device=device) for _ in pipes] if G > 1 else None

    def run_loop(ingest, steps, warmup):
        """warmup untimed steps, then `steps` timed steps bracketed by barrier + synchronize on both sides.
        Returns (seconds, faces, batch latencies ms, per-step (idx, dec, counts, source batch) host copies).  Steps are numbered
        0 .. warmup + steps - 1 across both phases (ring slots, pinned buffers and stream pairs follow that number)."""
        batch_ms, results = [], []
        uploads = {}
        total = warmup + steps
        group = []                          # embed groups: the steps whose crops wait for their shared forward

        def source(i):
            return (i % ingest.depth) % nbatch if ingest is not None else i % nbatch

        def finish(i, host, r, emb_rows, s_emb, t_in):
            """match + decision + asynchronous copies of one step's ids to pinned host memory; the step's pending entry"""
            idx, score = sharded.match(emb_rows)
            dec = gm.decide_device(idx, score, 0.4)
            host["idx"].copy_(idx.to(torch.int64), non_blocking=True)
            host["dec"].copy_(dec.to(torch.int32), non_blocking=True)
            host["counts"].copy_(r["counts"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(s_emb)
            return host, ev, (idx, dec, r, s_emb), t_in, source(i)

        def enqueue(i, flush):
            """One step, enqueued without any host synchronisation: fixed per-frame face slots, device-side counts,
            asynchronous copy of the ids / decisions / counts to pinned host memory.  Returns the pending entries that became
            complete with this step (one; with embed groups none, or the whole group's)."""
            t_in = time.perf_counter()
            host = pinned[i % len(pinned)]
            pi = (i // G) % len(pipes)
            s_det, s_emb = pipes[pi]
            with torch.cuda.stream(s_emb):
                kw = {}
                if G > 1:
                    kw["crops_out"] = group_crops[pi][len(group) * q_rows:(len(group) + 1) * q_rows]
                if ingest is not None:          # PCIe-inclusive variant: pinned host ring -> device on a copy stream
                    for j in range(i, min(i + 1 + args.ingest_ahead, total)):      # never past the loop's last step
                        if j not in uploads:
                            uploads[j] = ingest.upload(j)
                    frames, ready = uploads.pop(i)
                    r = app.detect_embed_slots(frames, det_stream=s_det, ready_event=ready, **kw)
                    ingest.release(i)           # the warp (last reader of the frames) is queued on this stream by now
                else:
                    r = app.detect_embed_slots(batches[i % nbatch], det_stream=s_det, **kw)
                if G == 1:
                    return [finish(i, host, r, r["normed_embedding"], s_emb, t_in)]
                group.append((i, host, r, t_in))
                if len(group) < G and not flush:
                    return []
                _, normed = app.embed_slots(group_crops[pi][:len(group) * q_rows])
                done = [finish(gi, gh, gr, normed[k * q_rows:(k + 1) * q_rows], s_emb, gt) for k, (gi, gh, gr, gt) in enumerate(group)]
                group.clear()
                return done

        def fetch(pending):
            """ids on the host = end of the step; faces = detected faces (slots beyond a frame's count are ignored)."""
            host, ev, _keep, t_in, src = pending
            ev.synchronize()
            batch_ms.append((time.perf_counter() - t_in) * 1e3)      # frames handed over -> ids on the host
            results.append((host["idx"].clone(), host["dec"].clone(), host["counts"].clone(), src))   # 3 KB, for the self-check
            return int(host["counts"].sum())

        def step(i, pending, flush):
            """Software pipeline of depth args.depth: step i is enqueued before step i-depth+1's ids are pulled to the
            host, so neither HIP stream waits for the Python driver between steps."""
            pending.extend(enqueue(i, flush))
            n = 0
            while len(pending) >= args.depth + (G - 1):
                n += fetch(pending.popleft())
            return n

        def drain(pending):
            n = 0
            while pending:
                n += fetch(pending.popleft())
            return n

        pending = deque()
        for i in range(warmup):
            step(i, pending, i == warmup - 1)
        drain(pending)
        sync()
        t0 = time.perf_counter()
        faces = 0
        batch_ms.clear(); results.clear()
        for i in range(warmup, total):
            faces += step(i, pending, i == total - 1)
        faces += drain(pending)                    # all K steps' ids are on the host inside the timed region
        sync()
        return time.perf_counter() - t0, faces, batch_ms, results

    from collections import deque

    def sync():
        if world > 1 or args.force_exchange:
            dist.barrier()
        torch.cuda.synchronize()

    dt, faces, batch_ms, results = run_loop(ingest, args.steps, args.warmup)
    tot = torch.tensor([dt, float(faces)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, faces = float(tmax[0]), float(tsum[1])

    # ---- self-check (outside the timed region): EVERY timed step's ids / decisions / counts must equal a sequential
    # single-stream run of the batch that step processed (cross-stream hazards are silent: DESIGN.md 4.7)
    expect = {}
    bad_step = -1
    plant_hit = plant_all = 0
    for k, (idx_h, dec_h, cnt_h, src) in enumerate(results):            # src: the batch step k used
        if src not in expect:
            r = app.detect_embed_slots(batches[src])
            idx2, score2 = sharded.match(r["normed_embedding"])
            dec2 = gm.decide_device(idx2, score2, 0.4)
            expect[src] = (idx2.cpu(), dec2.cpu(), r["counts"].cpu())
            if src == 0:                    # kept for the oracle check below (host copies of batch 0's sequential run)
                gpu0 = {kk: r[kk].cpu().numpy() for kk in ("counts", "bbox", "kps", "det_score", "embedding")}
                gpu0["idx"], gpu0["dec"] = idx2.cpu().numpy(), dec2.cpu().numpy()
        idx2, dec2, cnt2 = expect[src]
        pl = planted[src]
        plant_all += int((pl >= 0).sum())
        plant_hit += int(((pl >= 0) & (idx_h == pl) & (dec_h == 1)).sum())
        valid = (torch.arange(FACES_PER_FRAME)[None, :] < cnt2[:, None]).reshape(-1)      # empty slots are undefined
        if not (torch.equal(cnt2, cnt_h) and torch.equal(idx2[valid], idx_h[valid]) and torch.equal(dec2[valid], dec_h[valid])):
            bad_step = k if bad_step < 0 else bad_step      # no break: sharded.match above is a collective
    # the verdict is collective: a rank that left alone would strand its peers in the next barrier
    flag = torch.tensor([1.0 if bad_step >= 0 else 0.0], device=device)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if float(flag[0]) > 0:
        if bad_step >= 0:
            print(f"rank {rank}: pipelined step {bad_step} disagrees with its sequential re-run - results invalid",
                  file=sys.stderr, flush=True)
        if world > 1 or args.force_exchange:
            dist.destroy_process_group()
        sys.exit(3)

    # ---- per-stage times of one batch alone on this stream (outside the timed region; HIP events).  One untimed pass
    # first (this stream has not run the stages yet: its side streams are created on first use), then the median of 3
    def stages_once():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        sb, ss, sk, sc = app.det.detect_batch(batches[0])
        ev[1].record()
        crops0 = torch.empty((FRAMES * FACES_PER_FRAME, 112, 112, 8), dtype=torch.float16, device=device)
        sk = sk.contiguous()
        app.lib.fr_warp_affine_5pt_slots(_lib.ptr(batches[0]), FRAMES, H, W, _lib.ptr(sk), _lib.ptr(sc),
                                         FACES_PER_FRAME, 112, _lib.ptr(crops0), _lib.stream_ptr())
        emb0, nrm0 = app.rec.forward(crops0)
        ev[2].record()
        i0, s0 = sharded.match(nrm0)
        gm.decide_device(i0, s0, 0.4)
        ev[3].record()
        torch.cuda.synchronize()
        return [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]
    stages_once()
    if world > 1 or args.force_exchange:    # the exchange's four parts, bracketed by HIP events inside ShardedGalleryMatcher.match
        sharded.timing = []
    st3 = np.median(np.array([stages_once() for _ in range(3)]), axis=0)
    exchange_ms = sharded.exchange_ms()
    sharded.timing = None
    stage_ms = {"detect": round(float(st3[0]), 3), "align_embed": round(float(st3[1]), 3), "match": round(float(st3[2]), 3)}
    # which detector arithmetic the batches took, and how many crops its exact f32 R-/O-Net pass re-evaluated against the capacity of
    # its work lists (a crop dropped past the capacity would keep a split-precision threshold decision: the run fails on it)
    det_path = dict(app.det._tls.path)
    det_path["pconv1_mfma_levels"] = len(det_path["pconv1_mfma_levels"])
    lists = getattr(app.det, "_ro_lists", None) if det_path.get("split_ro") else None
    if lists:
        det_path["exact_crops"] = {"rnet": int(lists[0][0]), "onet": int(lists[1][0]), "list_caps": list(app.det.ro_list_cap)}
        det_path["exact_list_overflow"] = bool(int(lists[0][0]) > app.det.ro_list_cap[0] or int(lists[1][0]) > app.det.ro_list_cap[1])

    # ---- instrumented pass (outside the timed region): HIP events around every conv launch
    app.rec.profile = []
    r = app.detect_embed_slots(batches[0])
    torch.cuda.synchronize()
    per = {}
    for variant, flops, e0, e1 in app.rec.profile:
        d = per.setdefault(variant, [0, 0.0, 0.0])
        d[0] += 1; d[1] += flops; d[2] += e0.elapsed_time(e1) * 1e-3
    app.rec.profile = None
    dom = max(per, key=lambda k: per[k][2])
    calls, flops, secs = per[dom]
    achieved = flops / secs / 1e12
    # HBM traffic per launch of that kernel: PMC counters from a separate rocprofv3 pass (profiles/, see the
    # note inside the file for the gfx950 FETCH_SIZE correction); None when no record matches the kernel
    traffic = None
    busy = None             # matrix-pipe busy fraction of that kernel from the SQ counters (same file; a separate --pmc run)
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
        rec = pmc["kernels"]
        if pmc.get("_mfma_busy", {}).get("kernel") == dom:
            busy = {k: pmc["_mfma_busy"][k] for k in ("whole_kernel", "k_loop", "source")}
        hits = [v for k, v in rec.items() if k.startswith(dom + " ") and "hbm_bytes_per_launch_corrected" in v]
        if hits:            # the kernel serves several layer shapes: take the one with the most launches
            traffic = max(hits, key=lambda v: v.get("dispatches", 0))["hbm_bytes_per_launch_corrected"]
    except (OSError, KeyError, ValueError):
        pass
    is_f8 = dom.startswith("conv_stage14_f8") or (dom.startswith("conv_halo") and dom.endswith(", true, 8, 0>"))
    peak = MFMA_PEAK_TFLOPS_F8 if is_f8 else MFMA_PEAK_TFLOPS
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": f"{PMC_FILE}: rocprofv3 --pmc passes of tools/pmc_traffic.py (separate run; counters "
                                  "cannot be read from inside this process)" if traffic is not None else None,
                "mfma_pipe_busy_pmc": busy,
                "launches_per_step": calls, "avg_launch_us": round(secs / calls * 1e6, 2),
                "algorithmic_gflop_per_launch": round(flops / calls / 1e9, 3)}

    # ---- yardstick (outside the timed region): the vendor library's plain f16 GEMM of the dominant conv's own shape on
    # this box - M = pixels of the batch's 14x14 maps, N = 256 couts, K = 9 * 256 - no gather, no epilogue
    if world == 1 and not args.no_side and (dom.startswith("conv_stage14_kernel") or (dom.startswith("conv_halo_kernel<2, 13, 256") and not dom.endswith(", true, 8, 0>"))):
        gm_, gn_, gk_ = FRAMES * FACES_PER_FRAME * 196, 256, 2304
        ga = torch.randn((gm_, gk_), device=device, dtype=torch.float16)
        gb = torch.randn((gn_, gk_), device=device, dtype=torch.float16)
        for _ in range(3):
            torch.matmul(ga, gb.t())
        ge0, ge1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ge0.record()
        for _ in range(20):
            torch.matmul(ga, gb.t())
        ge1.record()
        torch.cuda.synchronize()
        gus = ge0.elapsed_time(ge1) / 20 * 1e3
        roofline["library_gemm_same_shape"] = {"what": f"torch.matmul f16 {gm_}x{gn_}x{gk_} (hipBLASLt): the plain GEMM of ONE of the stage's 14x14 convs, no gather / epilogue",
                                               "us": round(gus, 2), "tflops": round(2.0 * gm_ * gn_ * gk_ / gus / 1e6, 1)}
        del ga, gb

    # ---- side measurements for the other half of BASELINE's metric (rank 0, one GPU, default workload only)
    side = {}
    if world == 1 and args.workload == "C2" and not args.no_side:
        side["latency_c1_ms"] = c1_latency(app, device)      # (before the pinned ring below: 2.4 GB of page-locked host memory)
        if args.ingest == "resident":           # PCIe-inclusive rate: every step's 398 MB cross PCIe from a pinned ring
            from facerecognition_infrenceengine_amd.ingest import FrameIngest
            ing = FrameIngest(FRAMES, H, W, device, depth=args.depth + G + args.ingest_ahead)
            for k in range(ing.depth):
                ing.host_buffer(k)[...] = batches[k % nbatch].cpu().numpy()
            # the link alone: ONE step's frames (pinned) -> device, nothing else running - what value_pcie / value is to be read against
            torch.cuda.synchronize()
            ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            dst = torch.empty_like(batches[0])
            src_t = ing._host[0]                    # slot 0's page-locked buffer
            dst.copy_(src_t, non_blocking=True)
            ce[0].record()
            for _ in range(5):
                dst.copy_(src_t, non_blocking=True)
            ce[1].record()
            torch.cuda.synchronize()
            copy_ms = ce[0].elapsed_time(ce[1]) / 5
            step_bytes = FRAMES * H * W * 3
            side["pcie"] = {"bytes_per_step": step_bytes, "copy_alone_ms": round(copy_ms, 3),
                            "copy_alone_gbps": round(step_bytes / copy_ms / 1e6, 2),
                            "floor_ms": round(copy_ms, 3),
                            "floor_faces_per_s": round(faces / args.steps / copy_ms * 1e3, 1),
                            "pinned": bool(src_t.is_pinned()),
                            "note": "one step's frames, pinned host memory -> HBM, timed alone with HIP events (median-free mean of 5); "
                                    "floor_* = the step time / rate at which the link alone would bound the PCIe-inclusive pipeline"}
            del dst
            n_p = max(6, min(args.steps, 40))            # as many steps as the headline loop: the pipeline fill is a fixed cost
            dt_p, faces_p, _, _ = run_loop(ing, n_p, 4)
            side["value_pcie"] = round(faces_p / dt_p, 1)
            side["pcie"]["ms_per_step_pcie"] = round(dt_p / n_p * 1e3, 3)
            side["pcie"]["link_busy_frac"] = round(copy_ms / (dt_p / n_p * 1e3), 3)
            side["value_pcie_note"] = (f"{n_p} steps, frames uploaded from pinned host memory on a copy stream every step, each upload issued "
                                       f"{args.ingest_ahead} steps ahead of its step")
            del ing

    # ---- who ran where (N > 1 / the one-rank rehearsal): every rank's device index, PCI address, uuid; two ranks on one device fail
    ranks_rec = None
    if world > 1 or args.force_exchange:
        mine = {"rank": rank, "local_rank": local_rank, **rank_identity(device)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_rec = sorted(gathered, key=lambda d: d["rank"])
        clash = ranks_sharing_a_device(ranks_rec)
        if clash and not args.same_device:
            if rank == 0:
                print(f"ranks share a device {clash}: {ranks_rec}", file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(5)
    if rank == 0:
        out = {"metric": f"faces/sec end-to-end @{H}p", "value": round(faces / dt, 1), "unit": "faces/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "fp8" if args.workload == "C5" else "f16", "data": "synthetic",
               "config": {"workload": f"{args.workload}: {FRAMES}x{H}p synthetic frames/GPU, MTCNN full pyramid (caps "
                                      f"512/64/{FACES_PER_FRAME}), ArcFace r100 {'fp8 body convs' if args.workload == 'C5' else 'f16'} embed, {GALLERY_ROWS}-row cosine "
                                      "gallery (row-sharded over ranks)",
                          "frames_per_step_per_gpu": FRAMES, "ingest": args.ingest, "faces_per_step": faces / args.steps,
                          "steps_per_embed_forward": G,
                          "gallery_rows": GALLERY_ROWS, "gallery_scan": args.gallery, "weights": "seeded synthetic",
                          "parallelism": f"frame-shard x{world} + gallery row-shard"},
               # amortised: wall time / faces.  Batch latency: frames handed to the pipeline -> that batch's ids on the
               # host, with args.depth batches in flight (the throughput setting; --depth 1 --one-stream is the
               # latency setting); per face = batch latency / faces of a batch
               "per_face_latency_ms": round(dt / max(faces / world, 1) * 1e3, 4),
               "p50_batch_latency_ms": round(float(np.percentile(batch_ms, 50)), 3),
               "p95_batch_latency_ms": round(float(np.percentile(batch_ms, 95)), 3),
               "p50_face_latency_ms": round(float(np.percentile(batch_ms, 50)) / max(faces / world / args.steps, 1), 4),
               "stage_ms_alone": stage_ms, "detector_path": det_path, **({"exchange_ms": exchange_ms} if exchange_ms is not None else {}),
               **({"ranks": ranks_rec, "ranks_distinct_devices": not args.same_device} if ranks_rec is not None else {}), **side,
               "self_check": f"all {len(results)} timed steps == sequential single-stream re-run (ids, decisions, counts)",
               "planted_top1": {"faces": plant_all, "matched_own_row": plant_hit,
                                "note": "timed faces whose top-1 id is the gallery row planted for them (embedding + "
                                        "N(0, 0.02), renormalised) and is accepted at 0.4; rank 0's steps.  A CONSISTENCY figure: "
                                        "the rows are planted from the GPU's own embeddings, so it shows that every timed step "
                                        "reproduces them through the scan, not that they are right - oracle_check does that"},
               "roofline": roofline}
        fail = None
        if det_path.get("exact_list_overflow"):
            fail = "the detector's exact R-/O-Net work list overflowed its capacity (MTCNNHIP.ro_list_cap)"
        if world == 1 and not args.no_cpu_baseline:
            # CPU oracle leg on copies of the first frames of batch 0, against the gallery the GPU scans; its per-frame
            # results are ALSO the check of the GPU's results for those frames (oracle_check; a mismatch fails the run)
            nf = max(1, min(args.cpu_frames, FRAMES))
            frames_h = list(batches[0][:nf].cpu().numpy())
            G_h = gm.G.cpu().numpy()
            gallery = {str(i): G_h[i] for i in range(G_h.shape[0])}
            sweep, res0, memo = [], None, {}
            c1 = args.workload == "C1"          # C1 exactly - warm-up 3, median of 20, at 1 thread and at all cores
            for t in [int(v) for v in args.cpu_threads.split(",") if v.strip() != ""]:
                reps = args.cpu_reps if args.cpu_reps > 0 else (20 if c1 else 1)
                rec, res = cpu_baseline(frames_h, gallery, threads=t or None, match_memo=memo, warmups=3 if (c1 and reps > 1) else 0, reps=reps)
                sweep.append(rec)
                res0 = res0 or res
            out["cpu_baseline"] = max(sweep, key=lambda r: r["value"])
            out["cpu_baseline_sweep"] = [{"cores": r["cores"], "value": r["value"], "ms_per_pass": r["ms_per_pass"]} for r in sweep]
            out["oracle_check"] = oracle_check(res0, gpu0, 1e-3)
            if not out["oracle_check"]["ok"]:
                fail = (fail + "; " if fail else "") + "oracle_check failed: the GPU's results for batch 0 differ from the CPU oracle's"
        print(json.dumps(out), flush=True)
        if fail:
            print(fail, file=sys.stderr, flush=True)
            if world > 1 or args.force_exchange:
                dist.destroy_process_group()
            sys.exit(4)
    if world > 1 or args.force_exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
