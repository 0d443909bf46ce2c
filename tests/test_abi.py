"""The C-ABI library loads and exports every symbol include/frhip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "frhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from facerecognition_infrenceengine_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(cdll, s), f"{s} declared in frhip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound but not declared in frhip.h"
    lib = _lib.load()
    # the binding, the header and the built library agree on the ABI version (a stale .so is refused by _lib.load())
    macro = int(re.search(r"#define FR_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "frhip.h")).read()).group(1))
    assert lib.fr_version() == macro == _lib.ABI_VERSION


def test_invalid_arguments_return_error_not_crash():
    from facerecognition_infrenceengine_amd import _lib
    lib = _lib.load()
    with pytest.raises(_lib.FrError, match="D must be 512"):
        lib.fr_gallery_match_f32(None, None, 1, 10, 256, 0, None, None, None, 0, None, 0, None)
    with pytest.raises(_lib.FrError):
        lib.fr_l2norm_rows_f32(None, None, 4, 7, None)
    with pytest.raises(_lib.FrError, match="null view"):
        lib.fr_gallery_match_view_f32(None, None, None, 1, 10, 512, None, None, None, 0, None)
    with pytest.raises(_lib.FrError, match="D must be 512"):
        import ctypes as C
        lib.fr_gallery_match_view_f32(None, None, C.c_void_p(16), 1, 10, 128, None, None, None, 0, None)
    with pytest.raises(_lib.FrError, match="D must be 512"):
        lib.fr_gallery_update_rows_f32(None, None, None, 3, 64, 0, None)
    with pytest.raises(_lib.FrError, match="null pointer"):
        lib.fr_gallery_update_rows_f32(None, None, None, 3, 512, 0, None)
    assert lib.fr_gallery_update_rows_f32(None, None, None, 0, 512, 0, None) == 0        # nothing to do: no pointers read
    with pytest.raises(_lib.FrError, match="unknown layer"):
        import ctypes as C
        one = C.c_void_p(16)          # never dereferenced: the layer id is rejected first
        lib.fr_dconv_mfma_f32(99, one, one, one, None, one, 1, 8, 8, None, None, None, 0, 0, None, 0, None, None)
    with pytest.raises(_lib.FrError, match="bad argument"):
        lib.fr_conv_splitk_epilogue(None, 2, 10, 64, 7, 7, None, 0, None, None, None, None)
    # round-4 entries of the batch detector: argument checks come before any launch
    import ctypes as C
    one = C.c_void_p(16)
    with pytest.raises(_lib.FrError, match="mode must be 0 or 1"):
        lib.fr_pnet_conv1_band(2, one, 1, 64, 64, 32, 32, one, one, one, None, one, None, None, 0, None)
    with pytest.raises(_lib.FrError, match="mode 0 writes the split map"):
        lib.fr_pnet_conv1_band(0, one, 1, 64, 64, 32, 32, one, one, one, None, None, None, None, 0, None)
    with pytest.raises(_lib.FrError, match="mode 1 needs the f32 map and a tile list"):
        lib.fr_pnet_conv1_band(1, one, 1, 64, 64, 32, 32, one, one, one, one, None, None, None, 0, None)
    with pytest.raises(_lib.FrError, match="bad argument"):
        lib.fr_pnet_conv1_band(0, one, 1, 64, 64, 2, 32, one, one, one, None, one, None, None, 0, None)      # a level below 3 rows
    with pytest.raises(_lib.FrError, match="bad argument"):
        lib.fr_pnet_band_tiles(one, 1, 4, 40, one, one, None)                                                  # a conv1 map below 5 x 5
    assert lib.fr_pnet_band_tiles_count(0, 10, 10) == 0 and lib.fr_pnet_band_tiles_count(2, 17, 70) == 2 * 3 * 3


def test_call_list_entry_layout_and_slots():
    """fr_call (include/frhip.h) as ctypes sees it, and the 8-byte slots the recorder writes: pointers as they are, ints
    sign-extended, floats as their IEEE bits in the low word; an empty list is refused through the ABI."""
    import struct
    from facerecognition_infrenceengine_amd import _lib
    text = open(os.path.join(ROOT, "include", "frhip.h")).read()
    assert re.search(r"int32_t fn; int32_t nargs;\s*uint64_t a\[22\];", text)
    assert ctypes.sizeof(_lib.Call) == 8 + 22 * 8 and _lib.Call.a.offset == 8
    for name, fid in _lib.SEQ_FN.items():
        assert re.search(r"FR_FN_[A-Z0-9_]+ = %d\b" % fid, text), (name, fid)
        assert len(_lib.SIGNATURES[name][1]) <= 22
    assert _lib._slot(0.6, _lib._F) == struct.unpack("<I", struct.pack("<f", 0.6))[0]
    assert _lib._slot(-1, _lib._I) == 0xFFFFFFFFFFFFFFFF and _lib._slot(7, _lib._Z) == 7
    assert _lib._slot(None, _lib._P) == 0 and _lib._slot(ctypes.c_void_p(4096), _lib._P) == 4096
    lib = _lib.load()
    with pytest.raises(_lib.FrError, match="no calls"):
        lib.fr_detect_sequence(None, 0)
    bad = (_lib.Call * 1)()
    bad[0].fn = 99
    with pytest.raises(_lib.FrError, match="unknown function id"):
        lib.fr_detect_sequence(bad, 1)
    # an entry's nargs must be its entry point's arity: checked for the WHOLE list before anything is launched
    for name, fid in _lib.SEQ_FN.items():
        two = (_lib.Call * 2)()
        two[0].fn, two[0].nargs = 8, 2                    # a well-formed event record in front (never executed)
        two[1].fn, two[1].nargs = fid, len(_lib.SIGNATURES[name][1]) - 1
        with pytest.raises(_lib.FrError, match=r"call 1 .*arguments, expected %d" % len(_lib.SIGNATURES[name][1])):
            lib.fr_detect_sequence(two, 2)


def test_device_code_has_no_packed_f32_ops(tmp_path):
    """packed-f32 VALU results went stale beside another stream's conv kernels, silently.  The
    library is built with the feature off; this disassembles every gfx950 code object in it and checks."""
    import shutil
    import subprocess
    from facerecognition_infrenceengine_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "libfrhip.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    cos = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert cos, "no device code objects found in libfrhip.so"
    n_mfma = 0
    for f in cos:
        asm = subprocess.run([objdump, "-d", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        bad = re.findall(r"v_pk_(?:mul|add|fma)_f32", asm)
        assert not bad, f"{len(bad)} packed-f32 ops in {f}: build with NOPK (csrc/Makefile)"
        n_mfma += len(re.findall(r"v_mfma_f32_16x16x32[_a-z0-9]*f16", asm))
    assert n_mfma > 100                      # and it really is the library with the MFMA kernels in it
