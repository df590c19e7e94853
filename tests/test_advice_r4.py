"""Regression tests for the round-4 advisor findings that can be exercised without a GPU."""
import threading

import torch

from cleandiffuser_amd.engine import optim, runtime, train


def test_zeroed_gradient_mark_is_keyed_on_the_tensor_not_its_address():
    """FusedAdamW marks the gradients it zeroed in place as 'counts as None'.  A gradient somebody ELSE dropped and a backward pass
    re-created (same address from the caching allocator, version 0 again) is a different tensor and must count as a gradient."""
    p = torch.nn.Parameter(torch.ones(8))
    opt = optim.FusedAdamW([p], lr=1e-3)
    p.grad = torch.zeros(8)
    opt._mark_untouched([p], as_none=True)
    assert not opt._has_grad(p)                       # zeroed by the optimiser, untouched since
    p.grad.add_(1.0)                                  # a backward pass accumulates in place: version moves
    assert opt._has_grad(p)
    opt._mark_untouched([p], as_none=True)
    assert not opt._has_grad(p)
    fresh = torch.zeros(8)                            # model.zero_grad() / p.grad = None, then a fresh tensor with the marked version
    while fresh._version < p.grad._version:
        fresh.add_(0.0)
    assert fresh._version == p.grad._version
    p.grad = fresh
    assert opt._has_grad(p), "a fresh gradient tensor must never alias the mark of the one the optimiser zeroed"


def test_signature_scopes_of_two_threads_do_not_share_an_id():
    """signature_scope caches a module's weight signature under the scope's id: a second thread that enters a scope while the first
    is inside must get its own id, or it reads a signature cached before an optimiser / EMA step changed the weights."""
    net = torch.nn.Linear(4, 4)
    seen = {}
    inside, go_on = threading.Event(), threading.Event()

    def first():
        with runtime.signature_scope():
            seen["a"] = (runtime._sig_scope.id, runtime._signature(net))
            inside.set()
            go_on.wait(10)

    t = threading.Thread(target=first)
    t.start()
    assert inside.wait(10)
    with torch.no_grad():
        net.weight.add_(1.0)                          # (an optimiser step on the main thread while the other scope is open)
    with runtime.signature_scope():
        seen["b"] = (runtime._sig_scope.id, runtime._signature(net))
    go_on.set()
    t.join()
    assert seen["a"][0] != seen["b"][0] and seen["a"][1] != seen["b"][1]
    assert runtime._sig_scope.depth == 0


def test_a_slot_that_goes_from_none_to_a_tensor_enters_the_signature():
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(2, 2)
            self.register_buffer("late", None)

    m = M()
    s0 = runtime._signature_now(m)
    m.late = torch.ones(3)                            # same number of buffer slots, one more tensor
    s1 = runtime._signature_now(m)
    assert len(s1) == len(s0) + 1
    m.late = None
    assert len(runtime._signature_now(m)) == len(s0)


def test_native_training_graph_is_not_taken_for_shapes_its_groupnorm_backward_cannot_serve(monkeypatch):
    """train.supports(): GroupNorm groups of 6 / 12 channels (model_dim 48) or of more than 256 channels have no library backward for the gain / shift
    gradients -- such nets keep the ATen autograd path instead of raising inside loss.backward(); a frozen net on inputs that need no
    gradient is not a training forward at all.  (Device checks are stubbed: this is the host-side rule.)"""
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d

    class FakeX:
        is_cuda, dtype, requires_grad = True, torch.float32, False

    def as_cuda(net):
        for p in net.parameters():
            monkeypatch.setattr(type(p), "is_cuda", property(lambda self: True), raising=False)
        return net
    ok = as_cuda(JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], timestep_emb_type="positional", attention=False, kernel_size=5))
    assert train.supports(ok, FakeX())
    odd = JannerUNet1d(23, model_dim=48, emb_dim=32, dim_mult=[1, 2], timestep_emb_type="positional", attention=False, kernel_size=5)
    assert not train.supports(odd, FakeX())
    # (round 5: groups of up to 256 channels have a library backward -- ChiUNet1d's 1024 / 2048-channel levels; 512 do not)
    assert train._groupnorms_ok(torch.nn.Sequential(torch.nn.GroupNorm(8, 1024), torch.nn.GroupNorm(8, 2048)))
    assert not train._groupnorms_ok(torch.nn.Sequential(torch.nn.GroupNorm(8, 4096)))
    ok.requires_grad_(False)
    assert not train.supports(ok, FakeX())            # frozen net, input without requires_grad: the fused forward's business
    FakeX.requires_grad = True
    assert train.supports(ok, FakeX())
