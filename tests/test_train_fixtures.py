"""loss() / update() of the solver classes against fixtures the REAL reference produced (oracle/train_cases.py,
oracle/gen_train_golden.py): same seeded timestep / noise / label-dropout draws, same AdamW + EMA arithmetic.  CPU here; the ROCm
device run (draws replayed from the CPU generator) is tests/test_gpu_parity.py::test_loss_and_update_match_reference_fixture."""
import numpy as np
import pytest

from conftest import golden_path
from oracle import train_cases


@pytest.mark.parametrize("name", sorted(set(train_cases.SCENARIOS) - train_cases.HEAVY))
def test_loss_and_update_reproduce_the_reference(name):
    gold = np.load(golden_path("train_" + name))
    out = train_cases.run(name, "amd", "cpu")
    assert set(gold.files) == set(out)
    for k in gold.files:
        np.testing.assert_allclose(out[k], gold[k], rtol=2e-6, atol=2e-6, err_msg=f"{name}/{k}")
