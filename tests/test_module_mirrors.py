"""Module mirrors that no sampling fixture exercises, against outputs of the REAL reference (tests/golden/modules.npz, produced by
oracle/gen_module_golden.py): extra backbones, classifier networks, classifier wrappers (logp, input gradients, QGPO loss)."""
import numpy as np
import pytest
import torch

from conftest import golden_path
from oracle import gen_module_golden as G


@pytest.mark.parametrize("name", list(G.specs()))
def test_module_forward_matches_reference(name):
    gold = np.load(golden_path("modules"))
    net, args = G.build("cleandiffuser_amd", name)
    with torch.no_grad():
        y = net(*args).numpy()
    np.testing.assert_allclose(y, gold[name], rtol=2e-6, atol=2e-6)


def test_classifier_wrappers_match_reference():
    gold = np.load(golden_path("modules"))
    out = G.wrapper_outputs("cleandiffuser_amd")
    for k, v in out.items():
        np.testing.assert_allclose(v, gold[k], rtol=2e-5, atol=2e-6, err_msg=k)
