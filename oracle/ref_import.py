"""Import the real CleanDiffuser reference (read-only mount) -- build container only.

torchvision is absent here and only the (out-of-scope) image condition encoders need it, so empty stand-in
modules are registered first.  Nothing is written to /root/reference.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CDX_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "cleandiffuser"))


def import_reference():
    if not available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT} (expected on GPU boxes)")
    sys.dont_write_bytecode = True
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import cleandiffuser  # noqa: F401
    return cleandiffuser
