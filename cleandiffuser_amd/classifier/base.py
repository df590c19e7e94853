"""Classifier-guidance plug-in (contract: reference classifier/base.py:9-90): owns ``model`` / ``model_ema`` of a
``BaseNNClassifier``, ``logp(x, noise, c) -> (b, 1)``, ``gradients(x, noise, c) -> (logp, d logp.sum() / dx)``, an Adam optimiser
(default lr 2e-4, weight decay 1e-4), an EMA copy, and ``save/load`` of the ``{"model", "model_ema"}`` checkpoint.

``update`` on a ROCm device: native forward / backward nodes + ``FusedAdam`` (see the method).  ``gradients`` on a ROCm device is served by explicit forward + backward kernels when the network is one the engine knows
(engine/classifier_grad.py: HalfJannerUNet1d; engine/mlp_grad.py: MSEClassifier / QGPOClassifier over the MLP networks); everything else differentiates ``logp`` with autograd as the reference does.
"""
from copy import deepcopy
from typing import Optional

import torch

_DEFAULT_ADAM = {"lr": 2e-4, "weight_decay": 1e-4}


class BaseClassifier:
    def __init__(self, nn_classifier, ema_rate: float = 0.995, grad_clip_norm: Optional[float] = None,
                 optim_params: Optional[dict] = None, device: str = "cpu"):
        self.device = device
        self.ema_rate = ema_rate
        self.grad_clip_norm = grad_clip_norm
        self.model = nn_classifier.to(device)
        self.model_ema = deepcopy(self.model).eval()
        # (a torch.optim.Adam subclass: on a ROCm device its step is one multi-tensor launch of cdx_optim_f32, L2 decay as torch's)
        from ..engine.optim import FusedAdam
        self.optim = FusedAdam(self.model.parameters(), **(_DEFAULT_ADAM if optim_params is None else optim_params))

    # ---- what a concrete classifier defines ----
    def loss(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor):
        raise NotImplementedError

    def logp(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        """log p(c | x_t, noise level): x (b, *x_shape), noise (b,), c (b, *c_shape) -> (b, 1)."""
        raise NotImplementedError

    # ---- guidance signal ----
    def gradients(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        if x.is_cuda:
            from ..engine import classifier_grad
            native = classifier_grad.gradients(self, x, noise, c)        # None: not a network with hand-written backward kernels
            if native is None:
                from ..engine import mlp_grad                            # the MLP / QGPO energy classifiers: explicit GEMM backward
                native = mlp_grad.gradients(self, x, noise, c)
            if native is not None:
                return native
        x.requires_grad_()
        logp = self.logp(x, noise, c)
        (grad,) = torch.autograd.grad([logp.sum()], [x])
        x.detach()                                                       # (no-op, as in the reference: x keeps requires_grad)
        return logp.detach(), grad.detach()

    # ---- training ----
    def update(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor, update_ema: bool = True):
        """One training step (reference classifier/base.py:47-58): loss -> zero_grad -> backward -> [clip] -> Adam -> [EMA].  On a ROCm
        device with a network the native training path serves (HalfJannerUNet1d) forward + backward run on the library's kernels -- one
        HIP-graph replay once the first step passed the capturability probe -- and clip + Adam + EMA are at most three launches."""
        from ..engine import train
        from ..engine.optim import FusedAdam
        opt = self.optim
        fused = isinstance(opt, FusedAdam) and opt.native()
        g = train.graphed_classifier_step(self, x, noise, y) if fused else None
        if g is not None:
            opt.zero_grad()                               # (in place: the captured backward adds into these very tensors)
            loss = g.replay(x, (noise, y))
            for p in g.written:
                opt._gver.pop(id(p), None)                # a replay writes gradients without moving their version counters
        else:
            loss = self.loss(x, noise, y)
            opt.zero_grad()
            with train.grads_in_place(self.model.parameters()):
                loss.backward()
        grad_norm = None
        clip = self.grad_clip_norm if isinstance(self.grad_clip_norm, float) else None
        if fused:
            opt.step(max_norm=clip, ema=(self.model, self.model_ema, self.ema_rate) if update_ema else None)
            if clip:
                grad_norm = opt.last_grad_norm.item()
        else:
            if clip:
                grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip).item()
            opt.step()
            if update_ema:
                self.ema_update()
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def ema_update(self):
        from ..utils.misc import ema_update
        ema_update(self.model, self.model_ema, self.ema_rate)

    def train(self):
        self.model.train()

    def eval(self):
        for m in (self.model, self.model_ema):
            m.eval()

    # ---- checkpoints ----
    def save(self, path):
        torch.save({name: getattr(self, name).state_dict() for name in ("model", "model_ema")}, path)

    def load(self, path):
        ckpt = torch.load(path, map_location=self.device)
        for name in ("model", "model_ema"):
            getattr(self, name).load_state_dict(ckpt[name])
