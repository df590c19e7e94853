"""TEST INFRASTRUCTURE (see oracle/__init__.py).  A plain-torch interpreter of the ``cdx_step`` records exactly as the device
applies them (csrc/cdx_bigbatch.hip:solver_step_kernel, csrc/cdx_unet2.hip solver section): lets the CPU suite check that a
plan builder (engine/plan.py) reproduces the real reference's samples without a GPU.  The network is any callable."""
import torch


def run_plan(plan, net, x, *, predict_noise, prior=None, fix_mask=None, noise=(), x_min=None, x_max=None, cond=None,
             w_cfg=0.0):
    """x: (B, ...) initial state.  net(x_in, t (B,), cond|None) -> prediction.  Returns the final state."""
    b = x.shape[0]
    m = 0.0 if fix_mask is None else fix_mask
    prev = x_old = None
    draws = iter(noise)
    t_dtype = torch.long if plan.t_is_integer else torch.float32
    for st in plan.steps:
        t = torch.full((b,), st.t, dtype=t_dtype)
        edm = st.kind >= 5
        x_in = st.alpha * x if edm else x
        with torch.no_grad():
            if cond is not None and w_cfg not in (0.0, 1.0):
                p = w_cfg * net(x_in, t, cond) + (1.0 - w_cfg) * net(x_in, t, torch.zeros_like(cond))
            else:
                p = net(x_in, t, cond if (cond is not None and w_cfg != 0.0) else None)
        k = st.k
        if edm:
            d = k[0] * x + k[1] * p
            if x_min is not None:
                d = torch.maximum(d, x_min)
            if x_max is not None:
                d = torch.minimum(d, x_max)
            if st.kind == 7:                     # consistency model: mask first, then re-noise for the next level
                xn = d if fix_mask is None else d * (1.0 - m) + prior * m
                x = xn + k[3] * next(draws) if st.noise else xn
                continue
            s = (x - d) / k[2]
            if st.kind == 5:
                xn = x - s * k[3]
                if st.push:
                    prev, x_old = s, x
            else:
                xn = x_old - (prev + s) / 2.0 * k[3]
        else:
            al, sg = st.alpha, st.sigma
            if predict_noise:
                if x_max is not None:
                    p = torch.maximum(p, (x - al * x_max) / sg)
                if x_min is not None:
                    p = torch.minimum(p, (x - al * x_min) / sg)
                eps, xth = p, (x - sg * p) / al
            else:
                if x_min is not None:
                    p = torch.maximum(p, x_min)
                if x_max is not None:
                    p = torch.minimum(p, x_max)
                xth, eps = p, (x - al * p) / sg
            z = next(draws) if st.noise else None
            if st.kind >= 3:
                if st.kind == 3:
                    xn = k[0] * (x - k[1] * (p * (1.0 - m)))
                else:
                    xn = k[0] * (k[1] * x + k[2] * (p * (1.0 - m) + x * m))
                if z is not None:
                    xn = xn + k[3] * z
            elif st.kind == 0:
                xn = k[0] * (x - k[1] * eps) + k[2] * eps
                if z is not None:
                    xn = xn + k[3] * z
            elif st.kind == 1:
                xn = k[0] * ((x - k[1] * eps) / k[2]) + k[3] * eps
            else:
                if st.flags & 1:
                    eps, xth = eps * (1.0 - m), xth * (1.0 - m) + x * m
                v = xth if st.vsel & 1 else eps
                if st.vsel == 2:
                    v = k[3] * xth - k[4] * prev
                if st.vsel == 3:
                    v = k[3] * eps - k[4] * prev
                xn = k[0] * x - k[1] * v
                if z is not None:
                    xn = xn + k[2] * z
            if st.push:
                prev = eps if st.push == 2 else xth
        if fix_mask is not None:
            xn = xn * (1.0 - m) + prior * m
        x = xn
    return x
