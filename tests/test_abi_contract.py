"""CPU checks of the drop-in boundary: libcdx.so loads, exports every symbol include/cdx.h declares, the ctypes
mirrors have the C layout, csrc/cdx_ops.h matches engine/program.py, and argument validation fails loudly."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_libcdx()
    from cleandiffuser_amd.engine import runtime
    return runtime.load_library()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "cdx.h")).read()
    names = re.findall(r"^(?:int|const char\*)\s+(cdx_\w+)\s*\(", hdr, flags=re.M)
    assert set(names) >= {"cdx_abi_version", "cdx_last_error", "cdx_unet1d_run", "cdx_probe_mfma_layout"}
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cdx.h but not exported by libcdx.so"
    assert lib.cdx_abi_version() == int(re.search(r"#define CDX_ABI_VERSION (\d+)", hdr).group(1))


def test_ctypes_mirrors_have_c_layout(tmp_path):
    from cleandiffuser_amd.engine import runtime
    fields = [f for f, _ in runtime.CdxUnet1dLaunch._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "cdx.h"', 'int main(void){',
           'printf("%zu %zu\\n", sizeof(cdx_unet1d_launch), sizeof(cdx_step));']
    src += [f'printf("%zu\\n", offsetof(cdx_unet1d_launch, {f}));' for f in fields]
    src += ['return 0;}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(runtime.CdxUnet1dLaunch)
    assert int(out[1]) == ctypes.sizeof(runtime.CdxStep) == 48
    for f, off in zip(fields, out[2:]):
        assert getattr(runtime.CdxUnet1dLaunch, f).offset == int(off), f


def test_op_word_layout_matches_header():
    from cleandiffuser_amd.engine import program as P
    text = open(os.path.join(ROOT, "cleandiffuser_amd", "csrc", "cdx_ops.h")).read()
    defs = {k: v for k, v in re.findall(r"#define CDX_(\w+) (\d+)\b", text)}
    for name, val in defs.items():
        py = name if hasattr(P, name) else name.replace("W_", "W_", 1)
        assert hasattr(P, py), f"program.py lacks {py}"
        assert getattr(P, py) == int(val), (name, val, getattr(P, py))
    hdr = open(os.path.join(ROOT, "include", "cdx.h")).read()
    assert int(re.search(r"#define CDX_OP_WORDS (\d+)", hdr).group(1)) == P.OP_WORDS


def test_launch_validation_fails_loudly(lib):
    from cleandiffuser_amd.engine import runtime
    L = runtime.CdxUnet1dLaunch()
    assert lib.cdx_unet1d_run(ctypes.byref(L), None) == -1
    assert b"null" in lib.cdx_last_error()


def test_missing_library_is_a_hard_error(tmp_path):
    from cleandiffuser_amd.engine import runtime
    with pytest.raises(RuntimeError, match="native library not found"):
        runtime.load_library(str(tmp_path / "nope.so"))
