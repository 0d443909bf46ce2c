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
    def __init__(self, embedding_manager, processor=None, capture_factory=None, max_wait_s=0.005, reopen_after=20,
                 dead_after=200):
        """``processor``: a ``FaceRecognitionProcessor`` (built lazily from ``embedding_manager`` when None).
        ``max_wait_s``: how long a turn waits for a first frame before it looks at ``running`` again; a turn never
        waits for slow cameras once it holds a frame (latency stays that of the slowest *present* frame).
        ``reopen_after`` / ``dead_after``: consecutive failed reads before a capture is re-opened / given up."""
        self.embedding_manager = embedding_manager
        self.processor = processor
        self.capture_factory = capture_factory
        self.max_wait_s = max_wait_s
        self.reopen_after, self.dead_after = max(int(reopen_after), 1), max(int(dead_after), 1)
        self.latest = {}                       # source -> last processed frame (the default consumer's store)
        self._latest_lock = threading.Lock()
        self.running = False
        self.threads = []
        self.frame_queues, self.result_queue = {}, None
        self.stats = {"batches": 0, "frames": 0, "dropped_results": 0, "largest_batch": 0, "dead_sources": []}

    # ---- capture side (infrenceServer.py:573-601)
    def capture_frames(self, source, frame_queue):
        """The reference retries a failed read at once (:589-592) - in its own OS process.  Here capture is a thread
        that shares the interpreter with the ONE batching loop feeding the GPU, so a camera that drops (an RTSP
        disconnect makes read() fail immediately) must not spin: failed reads back off (10 ms doubling to 100 ms, one
        warning per 5 s), after ``reopen_after`` consecutive failures the capture is re-opened, and after
        ``dead_after`` the source is marked dead (``stats['dead_sources']``) and its thread ends."""
        cap = self.capture_factory(source)
        logger.info("Camera %s initialized", source)
        fails, delay, last_warn = 0, 0.01, 0.0
        try:
            while self.running:
                ok, frame = cap.read()
                if not ok:
                    if frame is None and getattr(cap, "exhausted", False):
                        break
                    fails += 1
                    now = time.monotonic()
                    if now - last_warn > 5.0:
                        logger.warning("Camera %s: read failed (%d in a row)", source, fails)
                        last_warn = now
                    if fails >= self.dead_after:
                        logger.error("Camera %s: %d failed reads in a row: giving up on this source", source, fails)
                        self.stats["dead_sources"].append(source)
                        break
                    if fails % self.reopen_after == 0:
                        try:
                            cap.release()
                            cap = self.capture_factory(source)
                            logger.info("Camera %s re-opened", source)
                        except Exception as e:                 # stays failed: the next reads back off again
                            logger.error("Camera %s: re-open failed: %s", source, e)
                    time.sleep(delay)
                    delay = min(delay * 2, 0.1)
                    continue
                fails, delay = 0, 0.01
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
        ``self.result_queue`` as ``(source, frame)``.  A consumer thread always drains that queue (the reference's
        cv2.imshow loop, :648-667): it calls ``display(source, frame)`` when given, otherwise it keeps the latest
        processed frame per source (``latest_frame(source)``) - so the unchanged ``POST /api/camera/start`` route,
        which passes no display, neither fills the 10-deep queue nor makes the GPU work for dropped results."""
        if self.running:
            return
        if self.capture_factory is None:
            raise RuntimeError("CameraManager needs capture_factory=<callable(source) -> capture object>")
        self.running = True
        sources = list(sources)
        self.frame_queues = {s: queue.Queue(maxsize=2) for s in sources}      # :629
        self.result_queue = queue.Queue(maxsize=10)                            # :630
        for s in sources:
            t = threading.Thread(target=self.capture_frames, args=(s, self.frame_queues[s]), daemon=True)
            t.start(); self.threads.append(t)
        t = threading.Thread(target=self.process_cameras, args=(sources, company_id), daemon=True)
        t.start(); self.threads.append(t)

        def keep_latest(source, frame):
            with self._latest_lock:
                self.latest[source] = frame

        sink = display if display is not None else keep_latest

        def consume():
            while self.running:
                try:
                    sink(*self.result_queue.get(timeout=1))
                except queue.Empty:
                    continue
                except Exception as e:
                    logger.error("Result consumer failed: %s", e)
        t = threading.Thread(target=consume, daemon=True)
        t.start(); self.threads.append(t)

    def latest_frame(self, source):
        """Last processed frame of ``source`` kept by the default consumer (None before the first result)."""
        with self._latest_lock:
            return self.latest.get(source)

    def stop_cameras(self):
        self.running = False
        for t in self.threads:
            t.join(timeout=5)
        self.threads.clear()
        logger.info("All camera threads stopped")
