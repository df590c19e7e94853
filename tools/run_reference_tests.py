"""Drop-in check (build container only): run the REFERENCE's own hot-path unit tests with ``cleandiffuser`` resolved to
``cleandiffuser_amd``.  Nothing is copied and nothing is written under /root/reference (no bytecode, no pytest cache).

    python tools/run_reference_tests.py [extra pytest args]
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

REF_TESTS = "/root/reference/tests"
HOT_PATH = ["test_janner_unet.py", "test_chi_unet.py", "test_chi_transformer.py", "test_dit.py", "test_dql_mlp.py",
            "test_idql_mlp.py", "test_mlps.py", "test_pearce_mlp.py", "test_pearce_transformer.py", "test_sfbc_unet.py",
            "test_all_nn_classifier.py", "test_all_classifier.py", "test_diffusion_sde.py"]


class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """import cleandiffuser[.x.y]  ->  the cleandiffuser_amd[.x.y] module object itself."""

    def find_spec(self, name, path=None, target=None):
        if name == "cleandiffuser" or name.startswith("cleandiffuser."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module("cleandiffuser_amd" + spec.name[len("cleandiffuser"):])

    def exec_module(self, module):
        pass


def main(argv):
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.meta_path.insert(0, _Alias())
    import pytest
    files = [os.path.join(REF_TESTS, f) for f in HOT_PATH]
    return pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", "/tmp", "-o", "python_files=test_*.py", *files, *argv])


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
