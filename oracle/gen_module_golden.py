"""Fixtures for module mirrors that no sampling case exercises (classifier networks / wrappers, extra backbones): forward
outputs -- and for the classifier wrappers logp + input gradients -- of the REAL reference on synthetic weights and inputs.
Run in the build container:  python -m oracle.gen_module_golden   ->  tests/golden/modules.npz.  TEST INFRASTRUCTURE."""
import numpy as np
import torch

from cleandiffuser_amd.utils import synth_array, synth_state_dict

B = 5


def specs():
    """name -> (namespace attr path, ctor kwargs, input builder).  Inputs are synthetic arrays keyed by the spec name."""
    def f(name, *shape):
        return torch.from_numpy(synth_array(f"mod/{name}", shape))

    def t_int(name):
        return torch.from_numpy((np.abs(synth_array(f"mod/{name}/t", (B,))) * 7).astype(np.int64) % 10)

    def t_float(name):
        return torch.from_numpy(np.abs(synth_array(f"mod/{name}/t", (B,))) * 0.3 + 0.05)
    return {
        "SfBCUNet": ("nn_diffusion.SfBCUNet", dict(act_dim=6, emb_dim=32, hidden_dims=[64, 32, 16]),
                     lambda: (f("sfbc/x", B, 6), t_float("sfbc"), f("sfbc/c", B, 32))),
        "DVInvMlp": ("nn_diffusion.DVInvMlp", dict(obs_dim=7, act_dim=3, emb_dim=16, hidden_dim=64),
                     lambda: (f("dv/x", B, 3), t_int("dv"), f("dv/c", B, 14))),
        "PearceTransformer": ("nn_diffusion.PearceTransformer", dict(act_dim=4, To=2, emb_dim=32, trans_emb_dim=16, nhead=4),
                              lambda: (f("pt/x", B, 4), t_int("pt"), f("pt/c", B, 2, 32))),
        "MLPNNClassifier": ("nn_classifier.MLPNNClassifier", dict(x_dim=6, out_dim=2, emb_dim=16, hidden_dims=[32, 32]),
                            lambda: (f("mlpc/x", B, 6), t_int("mlpc"))),
        "QGPONNClassifier": ("nn_classifier.QGPONNClassifier", dict(obs_dim=7, act_dim=3, emb_dim=16, hidden_dims=[32, 32],
                                                                    timestep_emb_type="untrainable_fourier"),
                             lambda: (f("qg/x", B, 3), t_float("qg"), f("qg/y", B, 7))),
        "HalfDiT1d": ("nn_classifier.HalfDiT1d", dict(in_dim=5, out_dim=1, emb_dim=16, d_model=32, n_heads=4, depth=2),
                      lambda: (f("hd/x", B, 8, 5), t_int("hd"), f("hd/c", B, 16))),
    }


def build(root_pkg, name):
    import importlib
    path, kw, make_inputs = specs()[name]
    mod, cls = path.split(".")
    net = getattr(importlib.import_module(f"{root_pkg}.{mod}"), cls, None)
    if net is None:                                    # classes the reference does not export from the package __init__
        sub = {"DVInvMlp": "dvinvmlp"}[cls]
        net = getattr(importlib.import_module(f"{root_pkg}.{mod}.{sub}"), cls)
    net = net(**kw)
    net.load_state_dict(synth_state_dict(net.state_dict(), 5))
    net.eval()
    return net, make_inputs()


def wrapper_outputs(root_pkg):
    """logp and d logp / d x of the classifier wrappers (autograd on CPU)."""
    import importlib
    C = importlib.import_module(f"{root_pkg}.classifier")
    out = {}
    net, (x, t) = build(root_pkg, "MLPNNClassifier")
    y = torch.from_numpy(synth_array("mod/mse/y", (B, 2)))
    clf = C.MSEClassifier(net, temperature=0.7)
    logp, grad = clf.gradients(x.clone(), t, y)
    out["MSEClassifier/logp"], out["MSEClassifier/grad"] = logp.numpy(), grad.numpy()
    net, (x, t, obs) = build(root_pkg, "QGPONNClassifier")
    clf = C.QGPOClassifier(net)
    logp, grad = clf.gradients(x.clone(), t, obs)
    out["QGPOClassifier/logp"], out["QGPOClassifier/grad"] = logp.numpy(), grad.numpy()
    k = 4
    xs = torch.from_numpy(synth_array("mod/qg/support", (B, k, 3)))
    soft = torch.softmax(torch.from_numpy(synth_array("mod/qg/q", (B, k, 1))), 1)
    loss, _ = clf.loss(xs, t, {"soft_label": soft, "obs": obs})
    out["QGPOClassifier/loss"] = np.float32(loss.item())
    return out


def main(path="tests/golden/modules.npz"):
    from .ref_import import import_reference
    import_reference()
    out = {}
    for name in specs():
        net, args = build("cleandiffuser", name)
        with torch.no_grad():
            out[name] = net(*args).numpy()
        print(f"{name:20s} out{out[name].shape} |y|max={np.abs(out[name]).max():.3f}")
    out.update(wrapper_outputs("cleandiffuser"))
    np.savez_compressed(path, **out)


if __name__ == "__main__":
    main()
