"""Model-pack ONNX import (no GPU): the graph reader and the IResNet data-flow mapping of onnx_import.py on files
written by tests/helpers/onnx_write.py from seeded state dicts, in both export styles."""
import numpy as np
import pytest
import torch

from facerecognition_infrenceengine_amd import onnx_import, weights
from oracle import nets
from tests.helpers.onnx_write import write_iresnet_onnx


@pytest.mark.parametrize("arch", ["r18", "r50"])
def test_unfolded_graph_gives_back_the_state_dict(tmp_path, arch):
    st = weights.synth_iresnet_state(arch, seed=7)
    path = tmp_path / "w.onnx"
    write_iresnet_onnx(path, {k: v.numpy() for k, v in st.items()}, arch, fold_bn=False)
    got, garch = onnx_import.iresnet_state_from_onnx(str(path))
    assert garch == arch
    assert set(got) == set(st)
    for k, v in st.items():
        assert got[k].dtype == np.float32 and np.array_equal(got[k].reshape(v.shape), v.numpy()), k


def test_folded_graph_is_the_same_network(tmp_path):
    """BatchNorms folded into the convs / the fc by the exporter: the state dict has conv biases instead of those BNs,
    and the oracle forward (which treats a missing BN as identity) gives the same embedding."""
    arch = "r18"
    st = weights.synth_iresnet_state(arch, seed=8)
    path = tmp_path / "folded.onnx"
    write_iresnet_onnx(path, {k: v.numpy() for k, v in st.items()}, arch, fold_bn=True)
    got = weights.load_state(str(path))                           # the model-pack entry point
    assert "layer1.0.bn1.weight" in got and "layer1.0.bn2.weight" not in got and "layer1.0.conv1.bias" in got
    assert "features.weight" not in got and "bn2.weight" in got
    x = torch.randn(2, 3, 112, 112, generator=torch.Generator().manual_seed(1))
    want = nets.iresnet_forward(st, x, nets.IRESNET_LAYERS[arch])
    have = nets.iresnet_forward(got, x, nets.IRESNET_LAYERS[arch])
    cos = torch.nn.functional.cosine_similarity(want, have).min().item()
    assert 1.0 - cos < 1e-6 and (want - have).abs().max().item() < 1e-3 * want.abs().max().item()


def test_rejects_a_graph_that_is_not_an_iresnet(tmp_path):
    from tests.helpers import onnx_write as ow
    w = ow._Writer()
    x = w.node("Relu", ["input.1"])
    graph = b"".join(ow._ld(1, n) for n in w.nodes) + ow._ld(11, ow._ld(1, b"input.1")) + ow._ld(12, ow._ld(1, x.encode()))
    path = tmp_path / "other.onnx"
    path.write_bytes(ow._vi(1, 7) + ow._ld(7, graph))
    with pytest.raises(ValueError, match="expected one Conv"):
        onnx_import.iresnet_state_from_onnx(str(path))


def test_tensor_decoder_accepts_default_data_location_and_int32_payloads():
    """TensorProto field 14 (data_location) = 0 is DEFAULT and legal when written explicitly; only 1 (EXTERNAL) is
    unsupported.  Field 5 (int32_data) carries int32 payloads and float16 payloads as uint16 bit patterns."""
    from tests.helpers import onnx_write as ow
    a = np.arange(6, dtype=np.float32).reshape(2, 3) - 2.5
    base = b"".join(ow._vi(1, d) for d in a.shape) + ow._ld(8, b"t")
    name, got = onnx_import._tensor(base + ow._vi(2, 1) + ow._ld(9, a.tobytes()) + ow._vi(14, 0))
    assert name == "t" and np.array_equal(got, a)
    with pytest.raises(ValueError, match="external"):
        onnx_import._tensor(base + ow._vi(2, 1) + ow._vi(14, 1))
    h = a.astype(np.float16)
    packed = b"".join(ow._varint(int(v)) for v in h.view(np.uint16).ravel())
    _, got16 = onnx_import._tensor(base + ow._vi(2, 10) + ow._ld(5, packed))
    assert got16.dtype == np.float16 and np.array_equal(got16, h)
    ints = np.array([[3, -1, 0], [7, -70000, 2]], dtype=np.int32)
    packed = b"".join(ow._varint(int(v)) for v in ints.ravel())
    _, goti = onnx_import._tensor(base + ow._vi(2, 6) + ow._ld(5, packed))
    assert goti.dtype == np.int32 and np.array_equal(goti, ints)


def test_pack_with_onnx_files_but_no_readable_recognition_net_raises(tmp_path):
    """A model directory that HOLDS .onnx files none of which maps onto an ArcFace IResNet must not fall back to
    synthetic recognition weights silently (it would recognise nobody); unreadable files of any kind (truncated
    protobuf: struct.error / IndexError / TypeError, not only ValueError) are skipped with their reason."""
    from facerecognition_infrenceengine_amd import _lib
    from facerecognition_infrenceengine_amd.face_analysis import FaceAnalysis
    d = tmp_path / "models" / "pack"
    d.mkdir(parents=True)
    (d / "det_10g.onnx").write_bytes(b"\x3a\x05\x0a\x03abc")            # a graph with one garbage node
    (d / "genderage.onnx").write_bytes(b"\x08")                        # truncated varint field
    app = FaceAnalysis(name="pack", root=str(tmp_path))
    with pytest.raises(_lib.FrError, match="none of its .onnx files"):
        app._load_states()
    # with a readable recognition network beside them the others are skipped and the pack loads
    st = weights.synth_iresnet_state("r18", seed=3)
    write_iresnet_onnx(d / "w600k_r18.onnx", {k: v.numpy() for k, v in st.items()}, "r18", fold_bn=False)
    with pytest.warns(UserWarning, match="MTCNN detector"):
        rec, det = app._load_states()
    assert app.arch == "r18" and np.array_equal(rec["conv1.weight"].numpy(), st["conv1.weight"].numpy())
