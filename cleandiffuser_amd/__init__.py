"""cleandiffuser_amd -- MI355X-native diffusion-policy sampling engine behind CleanDiffuser's
``DiffusionModel.sample()/.loss()`` + ``BaseNNDiffusion``/``BaseNNCondition`` plug-in surfaces.
See DESIGN.md (hot path, HBM/LDS layout, kernels) and INTEGRATION.md (drop-in recipe).
"""
__version__ = "0.1.0"


def install_as_cleandiffuser():
    """Register this package under the name ``cleandiffuser`` so reference pipelines import it unchanged."""
    import importlib
    import sys
    pkg = sys.modules[__name__]
    sys.modules.setdefault("cleandiffuser", pkg)
    for sub in ("utils", "nn_diffusion", "nn_condition", "diffusion", "classifier", "nn_classifier", "invdynamic", "dataset"):
        try:
            mod = importlib.import_module(f"{__name__}.{sub}")
        except ImportError:
            continue
        sys.modules.setdefault(f"cleandiffuser.{sub}", mod)
        for name, child in list(sys.modules.items()):
            if name.startswith(f"{__name__}.{sub}."):
                sys.modules.setdefault("cleandiffuser." + name[len(__name__) + 1:], child)
