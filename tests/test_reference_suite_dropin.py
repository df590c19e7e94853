"""Drop-in gate: the REFERENCE's own hot-path unit tests (its 51 backbone / classifier / solver tests) run unmodified against this
package, with ``cleandiffuser`` resolved to ``cleandiffuser_amd`` by an import alias (tools/run_reference_tests.py).  Build
container only -- the reference tree does not exist on the GPU box, so the test skips there."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests"), reason="reference tree not present on this box")
def test_reference_hot_path_tests_pass_against_this_package(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py")], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    n = int(r.stdout.rsplit(" passed", 1)[0].split()[-1])
    assert n >= 51, tail


def test_alias_installer_resolves_reference_import_paths(tmp_path):
    """INTEGRATION.md's one-line switch: after ``install_as_cleandiffuser()`` the reference's import paths (package, sub-packages and the
    per-file modules pipelines import from) resolve to this package.  In a subprocess: the alias must not leak into this test session."""
    code = ("import cleandiffuser_amd; cleandiffuser_amd.install_as_cleandiffuser()\n"
            "from cleandiffuser.diffusion import DiscreteDiffusionSDE, ContinuousDiffusionSDE\n"
            "from cleandiffuser.diffusion.diffusionsde import DiscreteDiffusionSDE as D2\n"
            "from cleandiffuser.nn_diffusion import JannerUNet1d, DiT1d, ChiUNet1d\n"
            "import cleandiffuser.nn_diffusion.jannerunet as j\n"
            "from cleandiffuser.nn_condition import IdentityCondition\n"
            "from cleandiffuser.classifier import CumRewClassifier\n"
            "from cleandiffuser.nn_classifier import HalfJannerUNet1d\n"
            "from cleandiffuser.utils import at_least_ndim\n"
            "assert D2 is DiscreteDiffusionSDE and j.__name__ == 'cleandiffuser_amd.nn_diffusion.jannerunet'\n"
            "print(DiscreteDiffusionSDE.__module__)\n")
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == "cleandiffuser_amd.diffusion.diffusionsde"


@pytest.mark.skipif(not os.path.isdir("/root/reference/pipelines"), reason="reference tree not present on this box")
def test_every_hot_path_import_of_the_reference_pipelines_resolves(tmp_path):
    """Every ``from cleandiffuser... import ...`` statement of the reference's pipelines, tutorials and tests, resolved against this
    package through the alias: what does not resolve must be outside SURVEY.md section 8 (datasets, environments, image encoders)."""
    code = r'''
import ast, os, sys, importlib
import cleandiffuser_amd; cleandiffuser_amd.install_as_cleandiffuser()
gaps = set()
for root, _, files in os.walk("/root/reference"):
    if root.startswith("/root/reference/cleandiffuser"):
        continue
    for f in files:
        if not f.endswith(".py"):
            continue
        try:
            tree = ast.parse(open(os.path.join(root, f)).read())
        except SyntaxError:
            continue
        for n in ast.walk(tree):
            if isinstance(n, ast.ImportFrom) and n.module and n.module.split(".")[0] == "cleandiffuser":
                try:
                    m = importlib.import_module(n.module)
                except ImportError:
                    gaps.add(n.module)
                    continue
                gaps.update(n.module + ":" + a.name for a in n.names if a.name != "*" and not hasattr(m, a.name))
print("\n".join(sorted(gaps)))
'''
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    gaps = [g for g in r.stdout.split() if g]
    # (round 6: the Decision-Veteran horizon critic -- a post-sampling transformer value function of the dv_* pipelines, SURVEY section 2
    #  row 6 "rest OUT OF SCOPE" -- is no longer mirrored: the round-5 copy was a transcription without a native path)
    out_of_scope = ("cleandiffuser.dataset.", "cleandiffuser.env", "cleandiffuser.nn_condition:MultiImageObsCondition",
                    "cleandiffuser.utils:DVHorizonCritic")
    assert all(g.startswith(out_of_scope) for g in gaps), [g for g in gaps if not g.startswith(out_of_scope)]
    # (round 4, SURVEY 8(f4): the two D4RL-MuJoCo datasets the Diffuser / DQL / IDQL pipelines train from are mirrored, with HBM-resident
    #  buffers -- their import statements resolve too)
    assert not any(g.startswith(("cleandiffuser.dataset.d4rl_mujoco_dataset:D4RLMuJoCoDataset", "cleandiffuser.dataset.d4rl_mujoco_dataset:D4RLMuJoCoTDDataset",
                                 "cleandiffuser.dataset.dataset_utils:loop_dataloader")) or g == "cleandiffuser.dataset.d4rl_mujoco_dataset"
                   for g in gaps), gaps
    # (round 5: every class of the reference's four d4rl_* dataset files is mirrored on the padded-episode store)
    assert not any(g.startswith("cleandiffuser.dataset.d4rl_") for g in gaps), [g for g in gaps if g.startswith("cleandiffuser.dataset.d4rl_")]


@pytest.mark.skipif(not os.path.isdir("/root/reference/cleandiffuser"), reason="reference tree not present on this box")
def test_overlay_alias_keeps_the_reference_for_everything_outside_the_hot_path(tmp_path):
    """``install_as_cleandiffuser(overlay=True)`` next to an installed reference: datasets / utils stay the reference's, the diffusion
    model, backbones, conditions and classifiers come from this package, the image encoders this package does not mirror remain
    importable -- and a reference-built condition module drives this package's solver to the reference's own samples."""
    code = r'''
import sys, numpy as np, torch
from oracle.ref_import import import_reference        # (registers the torchvision stand-ins this container needs)
import_reference()
from cleandiffuser.diffusion import DiscreteDiffusionSDE as RefSDE
from cleandiffuser.nn_condition import IdentityCondition as RefCond
import cleandiffuser_amd
cleandiffuser_amd.install_as_cleandiffuser(overlay=True)
import cleandiffuser
from cleandiffuser.diffusion import DiscreteDiffusionSDE
from cleandiffuser.diffusion.diffusionsde import ContinuousDiffusionSDE
from cleandiffuser.nn_diffusion import JannerUNet1d
from cleandiffuser.nn_condition import MultiImageObsCondition, IdentityCondition
from cleandiffuser.classifier import CumRewClassifier
from cleandiffuser.utils import report_parameters
from cleandiffuser.dataset.base_dataset import BaseDataset
assert DiscreteDiffusionSDE.__module__ == "cleandiffuser_amd.diffusion.diffusionsde" and DiscreteDiffusionSDE is not RefSDE
assert cleandiffuser.diffusion.__name__ == "cleandiffuser_amd.diffusion" and JannerUNet1d.__module__.startswith("cleandiffuser_amd.")
assert CumRewClassifier.__module__.startswith("cleandiffuser_amd.") and IdentityCondition is not RefCond
assert MultiImageObsCondition.__module__ == "cleandiffuser.nn_condition.multi_image_condition"
assert report_parameters.__module__ == "cleandiffuser.utils.utils" and BaseDataset.__module__ == "cleandiffuser.dataset.base_dataset"
# a condition module built by the REFERENCE inside this package's solver: same samples as the all-reference agent
from cleandiffuser_amd.utils import load_synth
torch.manual_seed(0)
outs = []
for sde, cond_cls in ((RefSDE, RefCond), (DiscreteDiffusionSDE, RefCond)):
    net = load_synth(JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=3), 3)
    agent = sde(net, cond_cls(dropout=0.0), diffusion_steps=10, predict_noise=True, device="cpu")
    agent.eval()
    torch.manual_seed(7)
    x, _ = agent.sample(torch.zeros(3, 8, 6), solver="ddim", n_samples=3, sample_steps=5, condition_cfg=torch.ones(3, 16), w_cfg=1.5)
    outs.append(x.numpy())
assert np.abs(outs[0] - outs[1]).max() < 1e-5
print("overlay ok")
'''
    env = dict(os.environ, PYTHONPATH=ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "overlay ok" in r.stdout, r.stderr[-3000:]
