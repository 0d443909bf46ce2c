"""Frame ingest: pinned host ring -> device, overlapped with compute.

The reference hands every camera frame to ``FaceAnalysis.get`` as a pageable NumPy array
(/root/reference/infrenceServer.py:573-601: ``cap.read()`` -> ``recognize_faces(frame, ...)``), i.e. one
synchronous host-to-device copy per frame.  Here the capture/decoder side writes BGR frames straight into
page-locked ring slots (``host_buffer``), ``upload`` starts an asynchronous copy of a whole batch on a
dedicated HIP copy stream and returns an event; the detector stream waits for that event only, so the copy of
batch i+1 runs under the kernels of batch i.  The HUD overlay stays on the CPU (out of scope).
"""
import io

import numpy as np
import torch

from . import _lib


def decode_image(data):
    """Encoded image bytes (JPEG / PNG ...) -> ``uint8 [H,W,3]`` BGR, or None when the bytes do not decode: the
    ``cv2.imdecode(np.frombuffer(blob, np.uint8), cv2.IMREAD_COLOR)`` of the enrolment worker
    (/root/reference/trainingServer.py:219-224), with Pillow standing in for OpenCV (absent from the image).  Host
    side by design (decode stays on the CPU); JPEG decoders differ in their IDCT rounding, so
    pixel values can differ from OpenCV's by a unit - the lossless formats decode identically."""
    try:
        from PIL import Image
        with Image.open(io.BytesIO(bytes(data))) as im:
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except Exception:                        # the reference logs "Failed to decode" and skips the image
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


class FrameIngest:
    """Ring of ``depth`` batch slots: pinned ``uint8 [N,H,W,3]`` host buffers + matching device buffers."""

    def __init__(self, n, h, w, device="cuda:0", depth=3):
        """One asynchronous copy of the whole batch per slot on one copy stream (round 4 measured frame groups and a second copy
        stream: no gain / -5 %, profiles/r04_ingest_ab.txt; the knobs are gone)."""
        _lib.require_gpu()
        self.device = torch.device(device)
        self.shape, self.depth = (int(n), int(h), int(w), 3), int(depth)
        self._host = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() for _ in range(self.depth)]
        self._dev = [torch.empty(self.shape, dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
        self._stream = torch.cuda.Stream(device=self.device)
        self._busy = [None] * self.depth          # event: last consumer of the slot's device buffer

    def host_buffer(self, slot):
        """NumPy view of the slot's pinned buffer: the decoder / capture thread writes frames here."""
        return self._host[slot % self.depth].numpy()

    def decode_into(self, slot, index, data):
        """Decode image bytes straight into frame ``index`` of the slot's pinned buffer (no intermediate pageable
        copy).  False when the bytes do not decode or the picture is not the ring's H x W."""
        img = decode_image(data)
        if img is None or img.shape != self.shape[1:]:
            return False
        self.host_buffer(slot)[index] = img
        return True

    def upload(self, slot):
        """Start the H2D copy of the slot; returns (device frames, event that fires when they have landed).
        The copy waits for the slot's previous consumer (see ``release``)."""
        k = slot % self.depth
        with torch.cuda.stream(self._stream):
            if self._busy[k] is not None:
                self._stream.wait_event(self._busy[k])
            self._dev[k].copy_(self._host[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(self._stream)
        return self._dev[k], ev

    def release(self, slot, stream=None):
        """Mark the slot's device buffer free once the work queued so far on ``stream`` (default: current) is done."""
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream(self.device))
        self._busy[slot % self.depth] = ev
