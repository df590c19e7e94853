"""Deterministic synthetic weights / inputs (there is no network for checkpoints).

``synth_state_dict`` fills every entry of a ``state_dict`` from a numpy PCG64 stream keyed by
(seed, crc32(parameter name)) -- independent of module construction order and of torch's RNG, so
the *reference* modules (oracle/gen_golden.py, run in the build container) and the modules of this
package (tests, bench, smoke; run anywhere) receive bit-identical, non-degenerate weights without
shipping multi-MB checkpoints.  Norm gains/biases are perturbed away from (1, 0) and zero-initialised
heads (DiT) become non-zero so that parity tests exercise every term.
"""
import zlib
from typing import Dict

import numpy as np
import torch


def _fill(name: str, shape, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    leaf = name.rsplit(".", 1)[-1]
    n = int(np.prod(shape)) if len(shape) else 1
    u = rng.uniform(-1.0, 1.0, size=n).astype(np.float64)
    if leaf == "freqs":                                   # Fourier embedding frequencies ~ scale 16
        out = 16.0 * u
    elif len(shape) >= 2:                                 # Linear / Conv / ConvTranspose kernels
        fan_in = int(np.prod(shape[1:]))
        out = u / np.sqrt(max(fan_in, 1)) * 1.7
    elif leaf in ("weight", "g"):                         # norm gain
        out = 1.0 + 0.25 * u
    else:                                                 # biases, norm shifts, misc 1-d
        out = 0.1 * u
    return out.astype(np.float32).reshape(shape)


def synth_state_dict(reference_sd: Dict[str, torch.Tensor], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Return a new state_dict with the same keys/shapes/dtypes, filled deterministically."""
    out = {}
    for name, t in reference_sd.items():
        if not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        out[name] = torch.from_numpy(_fill(name, tuple(t.shape), seed)).to(dtype=t.dtype)
        if name.endswith("running_var"):              # BatchNorm statistics must stay a valid variance
            out[name] = out[name].abs() + 0.5
    return out


def load_synth(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    module.load_state_dict(synth_state_dict(module.state_dict(), seed))
    return module


def synth_array(tag: str, shape, seed: int = 0, scale: float = 1.0, normal: bool = True) -> np.ndarray:
    """Deterministic input tensors (noise, priors, conditions) keyed by a tag."""
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(tag.encode())]))
    a = rng.standard_normal(size=shape) if normal else rng.uniform(-1, 1, size=shape)
    return (scale * a).astype(np.float32)
