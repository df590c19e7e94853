"""The weight layouts of a training step as strided-gather jobs (engine/train.py:_geom, cdx_relayout_f32 in include/cdx.h): the job of
every layout kind, evaluated with plain index arithmetic, is the tensor the ATen expression (train._aten_pack) builds -- for contiguous
weights and for row slices of a packed parameter.  CPU; the kernel itself runs in tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from cleandiffuser_amd.engine import train

KINDS3 = ["conv", "convt_bwd", "conv_bwd", "conv_s2_mid", "conv_s2_outer", "convt_even", "convt_odd"]


def gather(w: torch.Tensor, n, st, off):
    flat = w.detach().contiguous().view(-1) if w.is_contiguous() else None
    base = w.storage_offset()
    store = torch.as_strided(w.detach(), (w.untyped_storage().nbytes() // 4,), (1,), 0).numpy()
    i0, i1, i2 = np.meshgrid(np.arange(n[0]), np.arange(n[1]), np.arange(n[2]), indexing="ij")
    return store[base + off + i0 * st[0] + i1 * st[1] + i2 * st[2]].reshape(-1), flat


@pytest.mark.parametrize("kind", KINDS3 + ["linear_t"])
def test_every_layout_job_is_the_aten_expression(kind):
    g = torch.Generator().manual_seed(3)
    k = 3 if kind.startswith("conv_s2") else (4 if kind.startswith("convt") else 5)
    full = torch.randn(12, 7, generator=g) if kind == "linear_t" else torch.randn(12, 7, k, generator=g)
    for w in (full, full[4:10]):                          # a whole parameter, a row slice of one (in_proj_weight[d:])
        want = train._aten_pack(kind, w)
        shape, n, st, off = train._geom(kind, w)
        got, _ = gather(w, n, st, off)
        assert tuple(want.shape) == shape and np.array_equal(got, want.reshape(-1).numpy()), kind


def test_layouts_are_what_the_nodes_used_to_build():
    """The ATen expressions themselves against the pack helpers the forward-only executors use (engine/blocks.py)."""
    from cleandiffuser_amd.engine import blocks
    w = torch.randn(6, 5, 4)
    even, odd = blocks.pack_conv_transpose_k4s2p1(w)
    assert torch.equal(train._aten_pack("convt_even", w), even) and torch.equal(train._aten_pack("convt_odd", w), odd)
    assert torch.equal(train._aten_pack("conv", w), blocks.pack_conv(w))
