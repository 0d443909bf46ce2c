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
