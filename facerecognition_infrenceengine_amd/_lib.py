"""ctypes binding of libfrhip.so (include/frhip.h).  No CPU fallback: a missing library
or a failing call raises."""
import ctypes as C
import struct
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libfrhip.so")

_lock = threading.Lock()
_lib = None
ABI_VERSION = 102          # include/frhip.h FR_ABI_VERSION this binding was written against (tests/test_abi.py compares)


class FrError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
                ("bias", C.c_void_p), ("slope", C.c_void_p), ("residual", C.c_void_p),
                ("out_f32_partial", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("bias_mode", C.c_int), ("splitk", C.c_int),
                ("x2", C.c_void_p), ("C2", C.c_int)]


class ConvStep(C.Structure):
    _fields_ = [("kind", C.c_int), ("args", ConvArgs)]


class ConvF8Args(C.Structure):
    _fields_ = [("x8", C.c_void_p), ("w8", C.c_void_p), ("y16", C.c_void_p), ("y8", C.c_void_p),
                ("oscale", C.c_void_p), ("bias", C.c_void_p), ("slope", C.c_void_p), ("residual", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("bias_mode", C.c_int), ("y8_mul", C.c_float), ("y8_sub", C.c_void_p)]


class RolePtr(C.c_void_p):
    """A device pointer whose ROLE in a recorded detector call list is known where it is passed (``ptr(t, role=...)``):
    the recorder notes the role beside the slot, and a replay patches exactly the slots that were recorded with a
    role - the frame and the four result tensors - never a slot that merely holds the same address."""
    role = None


class PnetLevel(C.Structure):
    """One pyramid level of fr_pnet_finish_levels (include/frhip.h fr_pnet_level)."""
    _fields_ = [("x1", C.c_void_p), ("head", C.c_void_p), ("workspace", C.c_void_p), ("H1", C.c_int), ("W1", C.c_int),
                ("scale", C.c_float), ("boxes", C.c_void_p), ("scores", C.c_void_p), ("regs", C.c_void_p),
                ("counts", C.c_void_p), ("block_counts", C.c_void_p)]


class Call(C.Structure):
    """One recorded call of fr_detect_sequence: function id + arguments as 8-byte slots (include/frhip.h fr_call)."""
    _fields_ = [("fn", C.c_int32), ("nargs", C.c_int32), ("a", C.c_uint64 * 22)]


_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
# entry points fr_detect_sequence can replay (ids: include/frhip.h FR_FN_*)
SEQ_FN = {"fr_dconv_mfma_f32": 1, "fr_pnet23_split_f16": 2, "fr_pnet_candidates": 3, "fr_sort_nms": 4, "fr_box_refine": 5,
          "fr_crop_conv1_f32": 6, "fr_stage_select": 7}

# name -> (restype, argtypes); every symbol include/frhip.h declares
SIGNATURES = {
    "fr_version": (_I, []),
    "fr_last_error_string": (C.c_char_p, []),
    "fr_device_count": (_I, []),
    "fr_l2norm_rows_f32": (_I, [_P, _P, _I, _I, _P]),
    "fr_gallery_match_workspace": (_Z, [_I, _L]),
    "fr_gallery_match_f32": (_I, [_P, _P, _I, _L, _I, _L, _P, _P, _P, _Z, _P, _I, _P]),
    "fr_gallery_match_view_f32": (_I, [_P, _P, _P, _I, _L, _I, _P, _P, _P, _Z, _P]),
    "fr_gallery_update_rows_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fr_gallery_match_f16_workspace": (_Z, [_I, _L]),
    "fr_gallery_match_f16": (_I, [_P, _P, _P, _I, _L, _I, _L, _P, _P, _P, _Z, _P, _I, _P]),
    "fr_gallery_match_f8_workspace": (_Z, [_I, _L]),
    "fr_gallery_match_f8": (_I, [_P, _P, _P, _I, _L, _I, _L, _P, _P, _P, _Z, _P, _I, _P]),
    "fr_f32_to_f8": (_I, [_P, _P, _L, _P]),
    "fr_f32_to_f16": (_I, [_P, _P, _L, _P]),
    "fr_match_decide": (_I, [_P, _P, _I, _F, _F, _P, _P]),
    "fr_match_pack_candidates": (_I, [_P, _P, _I, _P, _P]),
    "fr_match_reduce_shards": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "fr_exchange_pack_queries": (_I, [_P, _I, _I, _I, _P, _P]),
    "fr_exchange_counts": (_I, [_P, _I, _I, _I, _P, _P]),
    "fr_gallery_first_above_f32": (_I, [_P, _P, _I, _L, _I, _F, _I, _L, _P, _P, _P, _Z, _P]),
    "fr_cosine_matrix_f32": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "fr_mean_rows_f32": (_I, [_P, _I, _I, _P, _P]),
    "fr_conv_nhwc_f16": (_I, [C.POINTER(ConvArgs), _P]),
    "fr_conv_sequence": (_I, [C.POINTER(ConvStep), _I, _P]),
    "fr_conv_inblock_f16": (_I, [C.POINTER(ConvArgs), _P]),
    "fr_conv_nhwc_f8": (_I, [C.POINTER(ConvF8Args), _P]),
    "fr_quantize_f16_f8": (_I, [_P, _P, _L, _F, _P]),
    "fr_quantize_f16_f8_centred": (_I, [_P, _P, _L, _I, _P, _F, _P]),
    "fr_gptq_round_e4m3": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "fr_conv_stage14_weight_bytes": (_Z, [_I]),
    "fr_conv_stage14_pack": (_I, [_P, _P, _P]),
    "fr_conv_stage14_f16": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "fr_conv_walk64_weight_bytes": (_Z, [_I]),
    "fr_conv_walk64_pack": (_I, [_P, _P, _I, _P]),
    "fr_conv_walk64_f16": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "fr_conv_stage28_weight_bytes": (_Z, [_I]),
    "fr_conv_stage28_pack": (_I, [_P, _P, _P]),
    "fr_conv_stage28_f16": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "fr_conv_stage14_f8_weight_bytes": (_Z, [_I]),
    "fr_conv_stage14_f8_param_floats": (_Z, []),
    "fr_conv_stage14_f8_pack": (_I, [_P, _P, _P]),
    "fr_conv_stage14_f8": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "fr_conv_splitk_epilogue": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "fr_fc_reduce_l2norm": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "fr_warp_affine_5pt": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "fr_warp_affine_5pt_slots": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "fr_pyramid_resize_norm": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "fr_dconv_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "fr_dconv_mfma_f32": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _I, _P, _P]),
    "fr_crop_conv1_f32": (_I, [_I, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    "fr_crop_conv1_split": (_I, [_I, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "fr_crop_conv1_list_f32": (_I, [_I, _P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    "fr_ro_conv2_split": (_I, [_I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P]),
    "fr_ro_gemm_weight_bytes": (_Z, [_I]),
    "fr_ro_gemm_pack": (_I, [_I, _P, _P, _P]),
    "fr_ro_gemm_split": (_I, [_I, _P, _P, _P, _P, _P, _I, _P, _I, _P]),
    "fr_ro_margin_list": (_I, [_P, _I, _P, _I, _I, _F, _F, _P, _P, _I, _P]),
    "fr_ro_scatter_rows": (_I, [_P, _P, _P, _I, _I, _P, _P]),
    "fr_pnet23_workspace_bytes": (_Z, [_I, _I, _I]),
    "fr_pnet23_split_f16": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P, _Z, _P]),
    "fr_maxpool_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fr_pnet_conv1_band": (_I, [_I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "fr_pnet_band_tiles_count": (_Z, [_I, _I, _I]),
    "fr_pnet_band_tiles": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "fr_pnet_finish_levels": (_I, [C.POINTER(PnetLevel), _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _F, _P, _P]),
    "fr_pnet_candidates": (_I, [_P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _F, _P]),
    "fr_sort_nms": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P, _P, _I, _P]),
    "fr_box_refine": (_I, [_P, _P, _I, _P, _I, _I, _I, _P]),
    "fr_crop_resize_norm": (_I, [_P, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "fr_stage_select": (_I, [_P, _P, _I, _P, _I, _I, _F, _P, _P, _P, _I, _P, _P, _P]),
    "fr_detect_sequence": (_I, [C.POINTER(Call), _I]),
}

_NOCHECK = ("fr_version", "fr_device_count")


def _slot(v, t):
    """a call argument as the 8-byte slot fr_detect_sequence reads"""
    if t is _F:
        return struct.unpack("<I", struct.pack("<f", float(v)))[0]
    if t is _P:
        if v is None:
            return 0
        return int(v.value or 0) if isinstance(v, C.c_void_p) else int(v)
    return int(v) & 0xFFFFFFFFFFFFFFFF


class Lib:
    """Thin checked wrapper: ``lib.fr_xxx(...)`` raises FrError on a negative return."""

    def __init__(self, cdll):
        self._c = cdll
        self._calls = {}
        self._rec = threading.local()          # .calls: a list while a thread records its detector calls (MTCNNHIP), else absent
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
            if res is _I and name not in _NOCHECK:
                self._calls[name] = self._checked(name, fn)
            else:
                self._calls[name] = fn

    def _checked(self, name, fn):
        err = self._c.fr_last_error_string

        fid, types, rec = SEQ_FN.get(name), fn.argtypes, self._rec

        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise FrError(f"{name} failed ({rc}): {err().decode()}")
            if getattr(rec, "calls", None) is not None:
                # a launch the call list cannot express (fr_detect_sequence replays SEQ_FN only): a replay would skip
                # it and the kernels behind it would read what an earlier frame left in the work tensors
                rec.invalid = name
            return rc

        def call_rec(*a):                       # noted down when this thread is recording: (id, slots, [(slot, role)])
            rc = fn(*a)
            if rc != 0:
                raise FrError(f"{name} failed ({rc}): {err().decode()}")
            calls = getattr(rec, "calls", None)
            if calls is not None:
                calls.append((fid, [_slot(v, t) for v, t in zip(a, types)],
                              [(i, v.role) for i, v in enumerate(a) if isinstance(v, RolePtr)]))
            return rc
        return call_rec if fid is not None else call

    def start_recording(self):
        self._rec.calls, self._rec.invalid = [], None

    def note(self, fid, *slots):
        """append a pseudo call (FR_FN_EVENT_RECORD = 8 / FR_FN_STREAM_WAIT = 9) to this thread's recording, if any"""
        calls = getattr(self._rec, "calls", None)
        if calls is not None:
            calls.append((fid, [int(v.value or 0) if isinstance(v, C.c_void_p) else int(v) for v in slots], []))

    def stop_recording(self):
        """The recorded calls, or None when nothing was recorded or the recording is unusable: an entry point outside
        SEQ_FN was called meanwhile (``recording_invalid()`` names it)."""
        calls, self._rec.calls = getattr(self._rec, "calls", None), None
        self._rec.last_invalid = getattr(self._rec, "invalid", None)
        self._rec.invalid = None
        return None if self._rec.last_invalid else calls

    def recording_invalid(self):
        """name of the entry point that made this thread's last recording unusable, or None"""
        return getattr(self._rec, "last_invalid", None)

    def __getattr__(self, name):
        try:
            return self.__dict__["_calls"][name]
        except KeyError:
            raise AttributeError(name)


def use_library(path):
    """Developer tools only (tools/stamp_*.py): bind the diagnostic twin ``libfrhip_debug.so`` (``make debug``)
    instead of the product library.  Must be called before the first ``load()``."""
    global LIB_PATH, _lib
    with _lock:
        if _lib is not None:
            raise FrError("use_library() must be called before the library is loaded")
        LIB_PATH = path


def load():
    """Load the in-tree libfrhip.so.  Raises FrError if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): import it first so
            # this process has ONE HIP runtime, shared by torch's allocator/streams and our kernels
            import torch  # noqa: F401
            if not os.path.exists(LIB_PATH):
                raise FrError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C {os.path.dirname(LIB_PATH)}`; there is no CPU fallback")
            lib = Lib(C.CDLL(LIB_PATH))
            have = lib.fr_version()
            if have != ABI_VERSION:     # a stale .so: argument structs / signatures of another shape
                raise FrError(f"{LIB_PATH} implements ABI version {have}, this binding needs {ABI_VERSION}: rebuild it "
                              f"(`make -C {os.path.dirname(LIB_PATH)}`)")
            _lib = lib
        return _lib


def ptr(t, role=None):
    """Device pointer of a torch tensor (or None); ``role``: see RolePtr.  The pointer does not keep the tensor alive: pass NAMED tensors.
    ``ptr(x[sel].contiguous())`` written inline frees the temporary as soon as the pointer is taken, and a second
    temporary built for the next argument of the same call can be handed the same block - its producer kernel is
    then queued BEFORE the consumer and overwrites the data (found the hard way: face_analysis.py compact_embed)."""
    if t is None:
        return None
    if role is None:
        return C.c_void_p(t.data_ptr())
    p = RolePtr(t.data_ptr())
    p.role = role
    return p


def stream_ptr():
    """The current torch HIP stream as a void*."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise FrError("no HIP device visible: this engine has no CPU path")
