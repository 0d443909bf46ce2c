"""Camera batcher: the reference's ``CameraManager`` (/root/reference/infrenceServer.py:565-679) re-shaped so that
N cameras feed ONE engine in batches.

Reference topology: per source one capture process -> ``Queue(maxsize=2)`` (frames dropped when full, :595-598) ->
one processing process that owns its own model and calls ``recognize_faces(frame, company_id)`` ONE FRAME AT A TIME
(:603-622) -> shared ``Queue(maxsize=10)`` -> display loop (:648-667).  N cameras = N model copies on device 0
and batch size 1 everywhere.

Here the capture side and the queues (sizes, drop-when-full) are kept, but ALL sources are drained by ONE batching
loop: each turn takes at most one frame per source, stacks them and runs ``FaceRecognitionProcessor.recognize_batch``
(one pass of the slot pipeline over the batch), then hands ``(source, processed_frame)`` pairs to the result queue in
source order - the same items the reference's per-camera processes produce.  Threads replace the reference's forked
processes: there is one engine to share, the GPU does the work, and the 30 s gallery sync thread keeps running
(in the reference the forked children never see a sync: threads do not survive fork).

Capture is injected (``capture_factory(source)`` -> object with ``read() -> (ok, frame)`` and ``release()``): the
reference uses ``cv2.VideoCapture`` (:575-584), which is not a dependency of this package.
"""
import logging
import queue
import threading
import time

logger = logging.getLogger(__name__)


class CameraManager:
    def __init__(self, embedding_manager, processor=None, capture_factory=None, max_wait_s=0.005):
        """``processor``: a ``FaceRecognitionProcessor`` (built lazily from ``embedding_manager`` when None).
        ``max_wait_s``: how long a turn waits for a first frame before it looks at ``running`` again; a turn never
        waits for slow cameras once it holds a frame (latency stays that of the slowest *present* frame)."""
        self.embedding_manager = embedding_manager
        self.processor = processor
        self.capture_factory = capture_factory
        self.max_wait_s = max_wait_s
        self.running = False
        self.threads = []
        self.frame_queues, self.result_queue = {}, None
        self.stats = {"batches": 0, "frames": 0, "dropped_results": 0, "largest_batch": 0}

    # ---- capture side (infrenceServer.py:573-601)
    def capture_frames(self, source, frame_queue):
        if self.capture_factory is None:
            raise RuntimeError("CameraManager needs capture_factory=<callable(source) -> capture object>")
        cap = self.capture_factory(source)
        logger.info("Camera %s initialized", source)
        try:
            while self.running:
                ok, frame = cap.read()
                if not ok:
                    if frame is None and getattr(cap, "exhausted", False):
                        break
                    continue
                try:
                    frame_queue.put_nowait(frame)          # non-blocking: skip the frame when the queue is full (:595-598)
                except queue.Full:
                    pass
        finally:
            cap.release()
            logger.info("Camera %s released", source)

    # ---- batching loop: replaces the N process_camera processes (infrenceServer.py:603-622)
    def take_batch(self, sources):
        """At most one frame per source, in ``sources`` order; blocks up to ``max_wait_s`` for the first one."""
        got = []
        for s in sources:
            try:
                got.append((s, self.frame_queues[s].get_nowait()))
            except queue.Empty:
                pass
        if not got:
            deadline = time.monotonic() + self.max_wait_s
            while not got and time.monotonic() < deadline and self.running:
                for s in sources:
                    try:
                        got.append((s, self.frame_queues[s].get_nowait()))
                    except queue.Empty:
                        pass
                if not got:
                    time.sleep(0.0005)
        return got

    def process_cameras(self, sources, company_id):
        if self.processor is None:
            from .processor import FaceRecognitionProcessor
            self.processor = FaceRecognitionProcessor(self.embedding_manager)
        while self.running:
            batch = self.take_batch(sources)
            if not batch:
                continue
            try:
                self.process_batch(batch, company_id)
            except Exception as e:                         # logged and survived, as the reference's loop (:620-621)
                logger.error("Error processing cameras %s: %s", [s for s, _ in batch], e)

    def process_batch(self, batch, company_id):
        """batch: [(source, frame)].  Frames of one size go through the engine together; results leave in batch order."""
        by_shape = {}
        for k, (_, f) in enumerate(batch):
            by_shape.setdefault(tuple(f.shape), []).append(k)
        results = [None] * len(batch)
        for ks in by_shape.values():
            res = self.processor.recognize_batch([batch[k][1] for k in ks], company_id)
            for j, k in enumerate(ks):
                results[k] = None if res is None else res[j]
        self.stats["batches"] += 1
        self.stats["frames"] += len(batch)
        self.stats["largest_batch"] = max(self.stats["largest_batch"], len(batch))
        for (source, frame), res in zip(batch, results):
            out = frame if res is None else self.processor.annotate(frame, res)
            try:
                self.result_queue.put_nowait((source, out))   # skip when the consumer is behind (:614-617)
            except queue.Full:
                self.stats["dropped_results"] += 1
        return results

    # ---- control (infrenceServer.py:624-679); called by the /api/camera/start|stop routes
    def start_cameras(self, sources, company_id, display=None):
        """Starts the capture threads and the batching loop and returns; processed frames arrive on
        ``self.result_queue`` as ``(source, frame)``.  ``display(source, frame)``, when given, is called from a
        consumer thread for every result (the reference's cv2.imshow loop, :648-667)."""
        if self.running:
            return
        self.running = True
        sources = list(sources)
        self.frame_queues = {s: queue.Queue(maxsize=2) for s in sources}      # :629
        self.result_queue = queue.Queue(maxsize=10)                            # :630
        for s in sources:
            t = threading.Thread(target=self.capture_frames, args=(s, self.frame_queues[s]), daemon=True)
            t.start(); self.threads.append(t)
        t = threading.Thread(target=self.process_cameras, args=(sources, company_id), daemon=True)
        t.start(); self.threads.append(t)
        if display is not None:
            def show():
                while self.running:
                    try:
                        display(*self.result_queue.get(timeout=1))
                    except queue.Empty:
                        continue
            t = threading.Thread(target=show, daemon=True)
            t.start(); self.threads.append(t)

    def stop_cameras(self):
        self.running = False
        for t in self.threads:
            t.join(timeout=5)
        self.threads.clear()
        logger.info("All camera threads stopped")
