"""cleandiffuser_amd -- MI355X-native diffusion-policy sampling engine behind CleanDiffuser's
``DiffusionModel.sample()/.loss()`` + ``BaseNNDiffusion``/``BaseNNCondition`` plug-in surfaces.
See DESIGN.md (hot path, HBM/LDS layout, kernels) and INTEGRATION.md (drop-in recipe).
"""
__version__ = "0.1.0"


HOT_PATH_PACKAGES = ("diffusion", "nn_diffusion", "nn_condition", "classifier", "nn_classifier", "invdynamic")


def install_as_cleandiffuser(overlay: bool = False):
    """Register this package under the name ``cleandiffuser`` so reference pipelines import it unchanged.

    ``overlay=False``: the whole name resolves here (code that touches only the diffusion model).

    ``overlay=True``: an installed reference package stays in place for everything outside the hot path -- ``cleandiffuser.dataset``,
    ``cleandiffuser.env``, ``cleandiffuser.utils`` -- and only the hot-path sub-packages (``HOT_PATH_PACKAGES``) are replaced by this
    package's; names a replaced sub-package lacks here (the torchvision image encoders of ``nn_condition``) keep pointing at the
    reference's classes.  A pipeline then runs unmodified: its data loading is the reference's, its model is this engine's."""
    import importlib
    import sys
    pkg = sys.modules[__name__]
    subs = ("utils", "nn_diffusion", "nn_condition", "diffusion", "classifier", "nn_classifier", "invdynamic", "dataset")
    ref = None
    if overlay:
        ref = importlib.import_module("cleandiffuser")
        if ref is pkg or getattr(ref, "__file__", None) == pkg.__file__:
            raise RuntimeError("install_as_cleandiffuser(overlay=True): no reference `cleandiffuser` package to overlay (the name "
                               "already resolves to cleandiffuser_amd)")
        subs = HOT_PATH_PACKAGES
    else:
        sys.modules.setdefault("cleandiffuser", pkg)
    for sub in subs:
        try:
            mod = importlib.import_module(f"{__name__}.{sub}")
        except ImportError:
            continue
        if overlay:
            try:
                theirs = importlib.import_module(f"cleandiffuser.{sub}")
            except ImportError:
                theirs = None
            for name in [n for n in sys.modules if n == f"cleandiffuser.{sub}" or n.startswith(f"cleandiffuser.{sub}.")]:
                del sys.modules[name]
            if theirs is not None and theirs is not mod:
                for k in dir(theirs):                      # e.g. MultiImageObsCondition: not mirrored here, still importable
                    if not k.startswith("_") and not hasattr(mod, k):
                        setattr(mod, k, getattr(theirs, k))
            sys.modules[f"cleandiffuser.{sub}"] = mod
            setattr(ref, sub, mod)
        else:
            sys.modules.setdefault(f"cleandiffuser.{sub}", mod)
        for name, child in list(sys.modules.items()):
            if name.startswith(f"{__name__}.{sub}."):
                alias = "cleandiffuser." + name[len(__name__) + 1:]
                if overlay:
                    sys.modules[alias] = child
                else:
                    sys.modules.setdefault(alias, child)
