"""Binding + launch helpers of the fused program kernel (``cdx_unet2_run`` / ``cdx_unet2_embtab``, include/cdx.h,
csrc/cdx_unet2.hip).  Same contract as runtime.py: PyTorch owns device memory and the stream, tensors cross the boundary as raw
pointers, no CPU / eager fallback -- a backbone the program compiler (program2.py) does not take is served by the GEMM
executors (bigbatch.py) or, failing that, is reported unsupported; nothing here computes in Python.

What the host keeps per (weights version, plan): the compiled program (ops, item tables, packed blob) and the FiLM table of the
plan's step records -- ``map_noise(t)`` -> ``cdx_unet2_embtab`` -- so a steady-state ``sample()`` call is ONE kernel launch and
no ATen launch.  The same launch record serves the U-Nets (Janner, Chi at small batch), the classifier's log-p pass, the
batch-tiled MLP programs and the small-batch mode that spreads one trajectory over several workgroups of one XCD
(DESIGN.md section 6).
"""
import ctypes
import os
import warnings
import weakref
from typing import Optional

import torch

from . import program2 as P2
from . import runtime as R


class CdxUnet2EmbtabArgs(ctypes.Structure):
    _fields_ = [("wblob", ctypes.c_void_p),
                ("emb_dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("md", ctypes.c_int32), ("n_emb", ctypes.c_int32),
                ("w0", ctypes.c_int32), ("b0", ctypes.c_int32), ("w2", ctypes.c_int32), ("b2", ctypes.c_int32),
                ("w3", ctypes.c_int32), ("b3", ctypes.c_int32),
                ("temb", ctypes.c_void_p), ("n_rows", ctypes.c_int32), ("out", ctypes.c_void_p),
                ("out_ld", ctypes.c_int32), ("col0", ctypes.c_int32),
                ("w4", ctypes.c_int32), ("b4", ctypes.c_int32), ("n_raw", ctypes.c_int32), ("col4", ctypes.c_int32)]


class CdxUnet2Launch(ctypes.Structure):
    _fields_ = [("ops", ctypes.c_void_p), ("wblob", ctypes.c_void_p),
                ("n_ops", ctypes.c_int32), ("traj_floats", ctypes.c_int32), ("traj_per_wg", ctypes.c_int32),
                ("n_waves", ctypes.c_int32), ("tune", ctypes.c_int32),
                ("x_off", ctypes.c_int32), ("x_stride", ctypes.c_int32),
                ("pred_off", ctypes.c_int32), ("pred_stride", ctypes.c_int32), ("prev_off", ctypes.c_int32),
                ("stage_off", ctypes.c_int32),
                ("batch", ctypes.c_int32), ("horizon", ctypes.c_int32), ("dim", ctypes.c_int32),
                ("traj_first", ctypes.c_int32), ("traj_count", ctypes.c_int32),
                ("emb", ctypes.c_void_p), ("emb_ld", ctypes.c_int32),
                ("steps", ctypes.c_void_p), ("n_steps", ctypes.c_int32), ("predict_noise", ctypes.c_int32),
                ("x_in", ctypes.c_void_p), ("prior", ctypes.c_void_p), ("fix_mask", ctypes.c_void_p),
                ("noise", ctypes.c_void_p), ("x_min", ctypes.c_void_p), ("x_max", ctypes.c_void_p),
                ("x_out", ctypes.c_void_p), ("init_blend", ctypes.c_int32), ("x_scale", ctypes.c_float),
                ("cg_scale", ctypes.c_void_p), ("grad_off", ctypes.c_int32), ("grad_stride", ctypes.c_int32),
                ("with_backward", ctypes.c_int32), ("ws", ctypes.c_void_p), ("ws_floats", ctypes.c_int32),
                ("compact", ctypes.c_int32), ("prof", ctypes.c_void_p),
                ("emb_per_traj", ctypes.c_int32), ("n_pass", ctypes.c_int32), ("emb_u", ctypes.c_void_p), ("cfg_w", ctypes.c_float),
                ("edm_plan", ctypes.c_int32), ("logp_out", ctypes.c_void_p), ("logp_first_op", ctypes.c_int32),
                ("logp_head_op", ctypes.c_int32), ("ctx", ctypes.c_void_p), ("mlp", ctypes.c_int32),
                ("split_k", ctypes.c_int32), ("xchg_floats", ctypes.c_int32), ("xbuf", ctypes.c_void_p), ("run_if", ctypes.c_void_p),
                ("xerr", ctypes.c_void_p), ("xseq0", ctypes.c_uint32), ("xtick0", ctypes.c_uint32), ("split_group", ctypes.c_int32),
                ("fault", ctypes.c_int32)]


_declared = False


def _lib():
    global _declared
    lib = R.load_library()
    if not _declared:
        lib.cdx_unet2_run.argtypes = [ctypes.POINTER(CdxUnet2Launch), ctypes.c_void_p]
        lib.cdx_unet2_run.restype = ctypes.c_int
        lib.cdx_unet2_embtab.argtypes = [ctypes.POINTER(CdxUnet2EmbtabArgs), ctypes.c_void_p]
        lib.cdx_unet2_embtab.restype = ctypes.c_int
        _declared = True
    return lib


class _Compiled2:
    def __init__(self, prog: Optional[P2.Program2], sig, why: Optional[str] = None):
        self.prog, self.sig, self.why = prog, sig, why
        self.ops_dev = None if prog is None else torch.from_numpy(prog.ops_buffer.copy()).to(prog.blob.device)


_cache = weakref.WeakKeyDictionary()


def enabled() -> bool:
    return os.environ.get("CDX_UNET2", "1") != "0"


def compiled2(module, horizon: int, nw: int = P2.NW2, compact: bool = False) -> _Compiled2:
    """The module's v2 program at this horizon for `nw` waves per workgroup (``.prog is None`` + ``.why`` when the v2 compiler
    does not take it).  `compact`: the small-LDS variant that lets three trajectories share a workgroup (state and multistep memory
    in global memory, in-place residual outputs, capped staging area) -- a third of 160 KiB or it does not exist."""
    per_mod = _cache.setdefault(module, {})
    sig = R._signature(module)
    key = (horizon, nw, bool(compact))
    hit = per_mod.get(key)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            kw = dict(compact=True, max_stage=COMPACT_STAGE, max_lds_bytes=(160 * 1024) // 3 // 16 * 16) if compact else {}
            lower = P2.compile_chiunet2 if R._is_chiunet(module) else P2.compile_janner2
            try:
                comp = _Compiled2(lower(module, horizon, nw=nw, **kw), sig)
            except ValueError:
                if compact or nw != P2.NW2_MAX or os.environ.get("CDX_UNET2_COMPACT_T1", "1") == "0":
                    raise
                # nets whose default plan does not fit 160 KiB (model_dim 64 at H = 64: the antmaze Diffuser, 193 KB) may still fit as
                # a compact program with the whole LDS to itself: one trajectory per workgroup
                comp = _Compiled2(lower(module, horizon, nw=nw, compact=True), sig)
        except ValueError as e:                  # the documented 'does not fit / unsupported layer' signal; invariant failures propagate
            comp = _Compiled2(None, sig, str(e))
    per_mod[key] = comp
    return comp


COMPACT_STAGE = 2304       # floats of staging area per op in a compact program


def _structural(module, horizon: int) -> Optional[str]:
    if not enabled():
        return "disabled by CDX_UNET2=0"
    if R._is_chiunet(module):
        if not module.obs_as_global_cond:
            return "ChiUNet1d with local conditioning has no v2 program"
    elif not R._is_janner(module):
        return f"{type(module).__name__} has no v2 program"
    n_down = sum(1 for lvl in module.downs if not isinstance(lvl[-1], torch.nn.Identity))
    if horizon % (1 << n_down) != 0:
        return f"horizon {horizon} not divisible by 2^{n_down}"
    return None


def supported(module, horizon: int) -> Optional[str]:
    """None when the v2 kernel runs `module` (JannerUNet1d, ChiUNet1d with a global condition) at `horizon`, else the reason."""
    why = _structural(module, horizon)
    if why is not None:
        return why
    first = compiled2(module, horizon, DEFAULT_NW)      # the shapes shape_for() tries, in its order
    return None if first.prog is not None else (compiled2(module, horizon, P2.NW2).why or first.why)


def compact_only(module, horizon: int) -> bool:
    """True when `module` runs on the v2 kernel only as a compact one-trajectory-per-workgroup program (default LDS plan too large):
    its sampling loops take the kernel below the GEMM executor's crossover batch; stand-alone forwards with per-sample timesteps
    and large batches stay with the implicit-GEMM executor."""
    comp = compiled2(module, horizon, DEFAULT_NW)
    return comp.prog is not None and comp.prog.compact


def film_table(comp: _Compiled2, module, t_vec: torch.Tensor, modules=None) -> torch.Tensor:
    """(rows, n_emb) FiLM table of the timesteps in `t_vec`: per network map_noise (the module's own embedding, a handful of ATen
    ops) and one cdx_unet2_embtab launch that fills the network's columns.  `modules`: the networks of a multi-network program
    (denoiser, classifier), default [module].  Callers memoise the result per plan."""
    prog = comp.prog
    dev = prog.blob.device
    mods = list(modules) if modules is not None else [module]
    assert len(mods) == len(prog.embtabs)
    out = torch.zeros((t_vec.shape[0], prog.n_emb), device=dev, dtype=torch.float32)
    for mod, e in zip(mods, prog.embtabs):
        with torch.no_grad():
            temb = R._f32c(mod.map_noise(t_vec), dev)
        args = CdxUnet2EmbtabArgs(wblob=prog.blob.data_ptr(), emb_dim=e["emb_dim"], hidden=e["hidden"], md=e["md"], n_emb=e["n_emb"],
                                  w0=e["w0"], b0=e["b0"], w2=e["w2"], b2=e["b2"], w3=e["w3"], b3=e["b3"],
                                  temb=temb.data_ptr(), n_rows=temb.shape[0], out=out.data_ptr(), out_ld=prog.n_emb, col0=e["col0"],
                                  w4=e["w4"], b4=e["b4"], n_raw=e["n_raw"], col4=e["col4"])
        R._check(_lib().cdx_unet2_embtab(ctypes.byref(args), R._stream_ptr(dev)), "cdx_unet2_embtab")
    return out


def plan_film_table(comp: _Compiled2, module, plan, device, modules=None, zero_row: bool = False) -> torch.Tensor:
    # one entry per (device, module, program variant): replaced -- not accumulated -- when the weight signature changes (a train /
    # evaluate loop that keeps the solver's cached plan would otherwise add a device table + a long tuple key per ema_update)
    memo = plan.__dict__.setdefault("_memo", {})
    key = ("film2", str(device), id(module), comp.prog.nw, bool(comp.prog.compact), comp.prog.ws_floats, len(comp.prog.embtabs), zero_row,
           comp.prog.meta.get("group_k", 0), comp.prog.meta.get("split_k", 0))        # (member programs: their own table, built from their own blob)
    hit = memo.get(key)
    if hit is None or hit[0] != comp.sig:
        t_vec = R.device_times(plan, device)
        if zero_row:            # one more row for timestep 0: the final log_p forward of a guided launch (reference diffusionsde.py:599)
            t_vec = torch.cat([t_vec, t_vec.new_zeros(1)])
        hit = memo[key] = (comp.sig, film_table(comp, module, t_vec, modules))
    return hit[1]


def traj_per_wg(prog: P2.Program2, batch: int) -> int:
    """Trajectories per workgroup: two as soon as the batch exceeds one workgroup per CU -- every streamed weight record then
    feeds twice the MFMAs (measured, B = 512: 7.05 ms against 8.0 ms for two co-resident single-trajectory workgroups per
    CU; B = 1024: 14.0 vs 16.1 ms).  CDX_UNET2_T forces 1 or 2."""
    forced = os.environ.get("CDX_UNET2_T")
    t = int(forced) if forced in ("1", "2") else (2 if batch > 256 else 1)
    if prog.lds_bytes(t) > 160 * 1024:
        t = 1
    return t


def n_waves(batch: int) -> int:
    """Waves per workgroup (the program is compiled per shape): 8 = two wave64 per SIMD.  CDX_UNET2_NW forces 4 or 8."""
    forced = os.environ.get("CDX_UNET2_NW")
    return int(forced) if forced in ("4", "8") else DEFAULT_NW


DEFAULT_NW = 8
DEFAULT_TUNE = 0


# Cost of one round of 256 workgroups with 1 / 2 / 3 trajectories each (config 2 on MI355X, measured: 4.41 / 5.63 / 7.66 ms); only
# the ratios matter -- they decide how a batch is cut into launches.
ROUND_COST = {1: 4.41, 2: 5.63, 3: 7.66}


# the same for the guided program of config 2 (B = 256: 10.1 ms; 13.4 ms per round of 512 at B = 3200; B = 768: 19.2 ms)
GUIDED_ROUND_COST = {1: 10.1, 2: 13.4, 3: 19.2}


def plan_parts(batch: int, tmax: int, round_cost=None):
    """Cut `batch` trajectories into launches [(first, count, trajectories per workgroup)]: ROUNDS of 256 workgroups, a round at T
    trajectories per workgroup costing ROUND_COST[T] whether it is full or not, the set of rounds with the smallest total (larger T
    first; ties: fewer launches).  B = 3200 is 12.5 rounds' worth at T = 1: three rounds at three per workgroup + two at two (13 x 256
    places, 34.2 ms by the table) beat four + a half-empty round at one (35.1 ms) -- CDX_UNET2_PLAN=bulk: the older rule, full rounds at
    the T that is cheapest per trajectory and ONE shape for the remainder."""
    ROUND_COST = round_cost or globals()["ROUND_COST"]
    if os.environ.get("CDX_UNET2_PLAN") == "bulk":
        best = None
        for tb in range(1, tmax + 1):
            per_round = N_CUS * tb
            rounds, parts = batch // per_round, []
            cost, bulk = rounds * ROUND_COST[tb], rounds * per_round
            if bulk:
                parts.append((0, bulk, tb))
            r = batch - bulk
            if r:
                tt = min(range(1, tmax + 1), key=lambda t: (-(-r // (N_CUS * t)) * ROUND_COST[t], t))
                cost += -(-r // (N_CUS * tt)) * ROUND_COST[tt]
                parts.append((bulk, r, tt))
            if best is None or cost < best[0] - 1e-9:
                best = (cost, parts)
        return best[1]
    best = None

    def walk(t: int, left: int, counts: tuple):
        nonlocal best
        if t == 0:
            if left > 0:
                return
            cost = sum(n * ROUND_COST[tt] for tt, n in counts)
            key = (round(cost, 6), sum(1 for _, n in counts if n), tuple(-n for _, n in counts))
            if best is None or key < best[0]:
                best = (key, counts)
            return
        for n in range(0, -(-max(left, 0) // (N_CUS * t)) + 1):
            walk(t - 1, left - n * N_CUS * t, counts + ((t, n),))
    # (the search is over the last few rounds only: everything before them runs at the T that is cheapest per trajectory)
    t_bulk = min(range(1, tmax + 1), key=lambda t: (ROUND_COST[t] / t, -t))
    n_bulk = max(0, batch // (N_CUS * t_bulk) - 4)
    walk(tmax, batch - n_bulk * N_CUS * t_bulk, ())
    parts, first = [], 0
    for t, n in ((t, n + (n_bulk if t == t_bulk else 0)) for t, n in best[1]):
        take = min(n * N_CUS * t, batch - first)
        if take > 0:
            parts.append((first, take, t))
            first += take
    return parts


def plan_for(module, horizon: int, batch: int):
    """(compiled program, launch parts) for an unguided request of `batch` trajectories.  CDX_UNET2_T forces one launch at 1, 2 or 3
    trajectories per workgroup; CDX_UNET2_SPLIT_TAIL=0 keeps a single launch."""
    comp, t = shape_for(module, horizon, batch)
    forced = os.environ.get("CDX_UNET2_T")
    three = compiled2(module, horizon, 8, compact=True) if (comp.prog is not None and comp.prog.nw == 8 and os.environ.get("CDX_UNET2_T3", "1") != "0") else None
    if three is not None and three.prog is None:
        three = None
    if forced == "3":
        return (three, [(0, batch, 3)]) if three is not None else (comp, [(0, batch, t)])
    if three is not None and os.environ.get("CDX_UNET2_COMPACT") == "1":          # test hook: the compact program at the forced shape
        return three, [(0, batch, int(forced) if forced in ("1", "2") else 1)]
    if forced in ("1", "2") or os.environ.get("CDX_UNET2_SPLIT_TAIL", "1") == "0" or R._prof["buf"] is not None:
        return comp, [(0, batch, t)]
    tmax = 3 if three is not None else (2 if comp.prog.lds_bytes(2) <= 160 * 1024 else 1)
    parts = plan_parts(batch, tmax)
    # every part of a launch runs the SAME program: the compact one (valid at 1..3 trajectories per workgroup) as soon as any part
    # asks for more trajectories than the default program's LDS plan holds (a cut into T = 2 parts is possible with tmax = 3 although
    # two default plans do not fit: model_dim 32, dim_mult [1,2,4] at H = 32 is 90.7 KB -- ADVICE r2)
    if any(comp.prog.lds_bytes(p[2]) > 160 * 1024 for p in parts) or max(p[2] for p in parts) == 3:
        assert three is not None
        return three, parts
    return comp, parts


def shape_for(module, horizon: int, batch: int):
    """(compiled program, trajectories per workgroup) the launch of `batch` trajectories uses.  The 8-wave shape stages more K
    slices, so its LDS plan is a little larger: when two of them do not fit next to each other but two of the 4-wave plan do,
    a batch that wants two trajectories per workgroup takes the 4-wave program."""
    nw = n_waves(batch)
    comp = compiled2(module, horizon, nw)
    if comp.prog is None and nw != P2.NW2:
        comp = compiled2(module, horizon, P2.NW2)
    t = traj_per_wg(comp.prog, batch)
    forced = os.environ.get("CDX_UNET2_T")
    want = int(forced) if forced in ("1", "2") else (2 if batch > 256 else 1)
    if t < want and comp.prog.nw != P2.NW2:
        alt = compiled2(module, horizon, P2.NW2)
        if alt.prog is not None and traj_per_wg(alt.prog, batch) == want:
            return alt, want
    return comp, t


def launch(comp: _Compiled2, *, batch, x_in, x_out, emb, steps_dev=None, n_steps=0, predict_noise=0, prior=None,
           fix_mask=None, noise=None, x_min=None, x_max=None, t_per_wg: Optional[int] = None, x_scale: Optional[float] = None,
           cg_scale=None, with_backward: bool = False, parts=None, emb_per_traj: bool = False, emb_u=None, cfg_w: float = 1.0,
           edm: bool = False, logp_out=None, ctx=None, split: int = 0, group: bool = False, run_if=None):
    """`run_if` (ordinary launches): the error word of the split / grouped launch enqueued just before on the same stream -- this launch
    is its REPAIR: it recomputes the request only if that word is set when it starts (include/cdx.h)."""
    if batch <= 0:
        return
    prog = comp.prog
    if "chi_film" in prog.meta and not emb_per_traj:
        raise ValueError("ChiUNet1d programs carry FiLM-scale ops: only the per-trajectory-table kernels decode them")
    mlp = "mlp" in prog.meta
    xbuf = xerr = None
    xseq0 = xtick0 = 0
    if split:
        assert split == prog.meta.get("group_k" if group else "split_k") and parts is None and t_per_wg in (None, 1)
        # A split / grouped launch ALWAYS has one workgroup per CU (256; 32 per XCD): the workgroups form their groups from per-XCD
        # tickets (HIP promises no workgroup -> XCD placement; include/cdx.h).  xbuf: a pair of exchange tiles per group, the ticket
        # counters (16 lines), who-ended-up-where records (256 words).
        n_grp = N_CUS // split
        assert (-(-batch // split) if group else batch) <= n_grp
        key = (x_in.device, R._stream_ptr(x_in.device))
        xf = prog.meta["xchg_floats"]
        need = n_grp * 4 * xf + 16 * 32 + N_CUS
        n_forwards = max(n_steps, 1)
        if "n_cut_ops" not in prog.meta:
            prog.meta["n_cut_ops"] = int(sum(1 for op in prog.ops if int(op[P2.W2_XG]) & P2.XG_XCHG))
        n_xchg = prog.meta["n_cut_ops"] * n_forwards
        st = _split_bufs.get(key)
        if st is None or st["layout"] != (n_grp, xf) or st["seq"] + n_xchg >= 2 ** 31 or st["tick"] >= 2 ** 31:
            # tiles of {value, sequence number} granules and ticket counters: zeroed ONCE per layout (and when a counter would wrap) --
            # sequence numbers and tickets keep increasing from launch to launch, so a granule left by an earlier launch never matches
            st = _split_bufs[key] = {"buf": torch.zeros(need, dtype=torch.float32, device=x_in.device), "seq": 0, "tick": 0,
                                     "layout": (n_grp, xf), "err": _split_err(x_in.device)}
        xbuf, xerr, xseq0, xtick0 = st["buf"], st["err"], st["seq"], st["tick"]
        # (the counters move only once the launch is enqueued -- below: a request the library rejects must not leave the host's idea of
        #  the ticket counters ahead of the device's)
    t = t_per_wg or traj_per_wg(prog, batch)
    prof = R._prof["buf"]
    # Two trajectories per workgroup fill the 256 CUs in rounds of 512 trajectories; a remainder of up to 256 is cheaper one
    # trajectory per workgroup (B = 3200: 6 full rounds + 128 trajectories: 4.6 instead of 5.9 ms for the last round).  Same stream,
    # same tensors, disjoint trajectory ranges; results do not depend on the split (T never changes a bit).
    given = parts
    parts = [(0, batch, t)]
    rounds = 2 * N_CUS
    if given is not None:
        parts = list(given)
    elif t == 2 and prof is None and os.environ.get("CDX_UNET2_SPLIT_TAIL", "1") != "0" and batch > rounds and 0 < batch % rounds <= N_CUS \
            and prog.lds_bytes(1) <= 160 * 1024:
        bulk = batch - batch % rounds
        parts = [(0, bulk, 2), (bulk, batch - bulk, 1)]
    ws, ws_floats = None, prog.ws_floats
    if emb_u is not None or edm:           # CFG pair / EDM kinds: [multistep memory or slope | x_old | conditional prediction] per trajectory
        assert cg_scale is None and not with_backward
        ws_floats = max(ws_floats, 3 * ((prog.horizon * prog.dim + 3) // 4 * 4))
    if ws_floats:                          # saved tensors of a two-trajectory guided program: scratch per (device, stream)
        key = (x_in.device, R._stream_ptr(x_in.device))
        ws = _ws.get(key)
        if ws is None or ws.numel() < (batch + 1) * ws_floats:            # + 1: spare block of the half-empty last workgroup
            _ws[key] = ws = torch.empty((batch + 1) * ws_floats, dtype=torch.float32, device=x_in.device)
    timing = R._timing
    if timing["on"]:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(torch.cuda.current_stream(x_in.device))
    for first, count, tp in parts:
        L = CdxUnet2Launch(
            ops=comp.ops_dev.data_ptr(), wblob=prog.blob.data_ptr(), n_ops=len(prog.ops), traj_floats=prog.traj_floats,
            traj_per_wg=tp, n_waves=prog.nw, tune=int(os.environ.get("CDX_UNET2_TUNE", DEFAULT_TUNE)), x_off=prog.x_off,
            x_stride=prog.x_stride, pred_off=prog.pred_off,
            pred_stride=prog.pred_stride, prev_off=prog.prev_off, stage_off=prog.stage_off,
            batch=batch, horizon=prog.horizon, dim=prog.dim, traj_first=first, traj_count=count, emb=emb.data_ptr(), emb_ld=emb.shape[1],
            steps=R._ptr(steps_dev), n_steps=n_steps, predict_noise=int(predict_noise),
            x_in=x_in.data_ptr(), prior=R._ptr(prior), fix_mask=R._ptr(fix_mask), noise=R._ptr(noise), x_min=R._ptr(x_min),
            x_max=R._ptr(x_max), x_out=x_out.data_ptr(), init_blend=0 if x_scale is None else 1,
            x_scale=1.0 if x_scale is None else float(x_scale), cg_scale=R._ptr(cg_scale), grad_off=prog.grad_off,
            grad_stride=prog.grad_stride, with_backward=int(with_backward or cg_scale is not None), ws=R._ptr(ws),
            ws_floats=ws_floats, compact=int(prog.compact), prof=R._ptr(prof), emb_per_traj=int(emb_per_traj),
            n_pass=2 if emb_u is not None else 1, emb_u=R._ptr(emb_u), cfg_w=float(cfg_w), edm_plan=int(edm),
            logp_out=R._ptr(logp_out), logp_first_op=prog.meta.get("cls_first", 0) if logp_out is not None else 0,
            logp_head_op=prog.meta.get("head_op", 0) if logp_out is not None else 0, ctx=R._ptr(ctx), mlp=int(mlp),
            split_k=int(split), xchg_floats=prog.meta.get("xchg_floats", 0) if split else 0, xbuf=R._ptr(xbuf), run_if=R._ptr(run_if),
            xerr=R._ptr(xerr), xseq0=int(xseq0), xtick0=int(xtick0), split_group=int(bool(group)),
            fault=int(os.environ.get("CDX_UNET2_FAULT", "0")) if split else 0)
        R._check(_lib().cdx_unet2_run(ctypes.byref(L), R._stream_ptr(x_in.device)), "cdx_unet2_run")
        if split:
            st["seq"] += n_xchg
            st["tick"] += N_CUS // 8
            st["last"] = (n_grp, xf, split, (xseq0 + 1) & 0x0fffffff)
    if timing["on"]:
        end.record(torch.cuda.current_stream(x_in.device))
        timing["repair_events" if run_if is not None else "events"].append((start, end))     # (an idle repair launch is not a kernel run)


N_CUS = 256          # the part these launch shapes are cut for (MI355X: 8 XCDs x 32 CUs); the split / grouped modes REQUIRE it (whole_chip)
_ws = {}
_split_bufs = {}     # (device, stream) -> {exchange tiles, sequence numbers handed out so far, error word} of the split programs


class CdxDeviceProps(ctypes.Structure):
    _fields_ = [("cu_count", ctypes.c_int32), ("xcc_count", ctypes.c_int32), ("lds_bytes_per_cu", ctypes.c_int32),
                ("wavefront", ctypes.c_int32), ("arch", ctypes.c_char * 32)]


_props = {}


def device_props(device) -> dict:
    """What the library found out about `device` (cdx_device_query, once per device: compute units and architecture from hipDeviceProp_t,
    the number of XCDs MEASURED by a probe launch that reads HW_REG_XCC_ID)."""
    device = torch.device(device)
    hit = _props.get(device)
    if hit is None:
        lib = _lib()
        lib.cdx_device_query.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(CdxDeviceProps)]
        lib.cdx_device_query.restype = ctypes.c_int
        out = CdxDeviceProps()
        scratch = torch.zeros(1, dtype=torch.int32, device=device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        R._check(lib.cdx_device_query(idx, scratch.data_ptr(), R._stream_ptr(device), ctypes.byref(out)), "cdx_device_query")
        hit = _props[device] = {"cu_count": int(out.cu_count), "xcc_count": int(out.xcc_count), "lds_bytes_per_cu": int(out.lds_bytes_per_cu),
                                "wavefront": int(out.wavefront), "arch": out.arch.decode()}
    return hit


def whole_chip(device) -> bool:
    """May the split / grouped programs run on `device`?  They launch exactly one workgroup per compute unit of a WHOLE MI355X -- 256
    CUs behind 8 XCDs, 32 workgroups per L2 -- and form their groups from per-XCD tickets; a CPX / NPS partition (32 CUs, one XCD), a CU
    mask or another part takes the ordinary programs instead of finding out through a lost granule.
    CDX_UNET2_ASSUME_WHOLE_CHIP=1 skips the query (tests of the refusal itself use =0)."""
    forced = os.environ.get("CDX_UNET2_ASSUME_WHOLE_CHIP")
    if forced in ("0", "1"):
        return forced == "1"
    p = device_props(device)
    return p["cu_count"] == N_CUS and p["xcc_count"] == 8 and p["arch"] == "gfx950" and p["wavefront"] == 64


last_exchange_error = {}    # device -> what the first member that gave up reported (diagnostics; see check_split_errors)
_split_errs = {}     # device -> one int32 in PINNED HOST memory: a member that loses a granule writes 1 there (over PCIe, on failure only)


def _split_err(device) -> torch.Tensor:
    t = _split_errs.get(device)
    if t is None:
        word = torch.zeros(8, dtype=torch.int32).pin_memory()      # [0] failed; [1..7] the first report (include/cdx.h: xerr)
        t = _split_errs[device] = (word, word.numpy())             # (the numpy view: reading it dispatches no ATen op)
    return t[0]


def _report_of(word) -> dict:
    what, wg, seq, item, xcc, member, grp = (int(v) for v in word[1:8])
    return {"what": "granule" if what == 1 else "ticket", "workgroup": wg, "sequence": seq, "item": item, "xcc": xcc, "member": member,
            "group": grp}


def note_exchange_failure(device) -> bool:
    """Has a split / grouped launch on `device` reported a lost granule (so far -- no synchronisation)?  If so both modes go off for the
    device for the rest of the process (its workgroups are evidently not co-resident behind one L2) and a warning says why.  The error
    word is NOT cleared here: repair launches that are still queued behind earlier split / grouped launches read it (cdx.h: run_if), and
    with the modes off nothing will raise it again.  Looking costs nothing (a numpy view of pinned host memory)."""
    ent = _split_errs.get(device)
    if ent is None or int(ent[1][0]) == 0:
        return False
    if _split_ok.get(device) is not False or _group_ok.get(device) is not False:
        _split_ok[device] = _group_ok[device] = _gguided_ok[device] = _sguided_ok[device] = False
        last_exchange_error[device] = _report_of(ent[1])
        warnings.warn("cdx_unet2_run (split / grouped program): a member never received a granule; the affected request was recomputed by "
                      "the ordinary program on the same stream (repair launch), and both modes are now off for this device "
                      f"(first report: {last_exchange_error[device]})")
    return True


def check_split_errors(device=None, wait: bool = True):
    """Raise if a split / grouped launch lost a granule (its polls are bounded: the launch ends; the workgroup that gave up stored NaN
    instead of its trajectories -- which the REPAIR launch behind it then overwrote with the ordinary program's result, so a caller of
    sample() never sees them).  `wait=True` (tests, explicit calls) synchronises the device first and clears the error word -- nothing
    is in flight then; `wait=False` only looks.  A device on which this fires loses both modes for the rest of the process."""
    for dev, (_, word) in list(_split_errs.items()):
        if device is not None and dev != device:
            continue
        if wait:
            torch.cuda.synchronize(dev)
        if int(word[0]) != 0:
            _split_ok[dev] = _group_ok[dev] = _gguided_ok[dev] = _sguided_ok[dev] = False
            last_exchange_error[dev] = _report_of(word)
            if wait:
                word[:] = 0
            raise RuntimeError("cdx_unet2_run (split / grouped program): a member never received a granule; the trajectories of that "
                               "launch were recomputed by the repair launch behind it, and both modes are now off for this device "
                               f"(first report: {last_exchange_error[dev]})")


def group_placement(device=None) -> dict:
    """Who ended up where in the LAST split / grouped launch on the current stream (synchronises): `workgroups` that drew a ticket,
    `groups` = complete groups (all k members recorded), `one_xcd_per_group`: every group's members report the same XCC id -- true by
    construction (a workgroup draws its ticket from the counter of the XCD it runs on), `per_xcd`: workgroups per XCC id."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    st = _split_bufs.get((device, R._stream_ptr(device)))
    if st is None or "last" not in st:
        return {"workgroups": 0, "groups": 0, "one_xcd_per_group": True, "per_xcd": {}}
    torch.cuda.synchronize(device)
    n_grp, xf, k, tag = st["last"]
    off = n_grp * 4 * xf + 16 * 32
    rec = (st["buf"][off: off + N_CUS].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff).reshape(n_grp, k)     # [group][member]
    live = ((rec >> 4) & 0x0fffffff) == tag
    xcc = rec & 15
    full = live.all(axis=1)
    same = bool(all((xcc[g] == xcc[g][0]).all() for g in range(n_grp) if full[g]))
    per = {}
    for v in xcc[live].tolist():
        per[int(v)] = per.get(int(v), 0) + 1
    return {"workgroups": int(live.sum()), "groups": int(full.sum()), "one_xcd_per_group": same, "per_xcd": per}


def split_factor(batch: int) -> int:
    """Workgroups per trajectory of an unconditional U-Net loop: 4 up to 64 trajectories, 2 up to 128 (every workgroup of the launch
    must be resident: ceil(B / 8) * 8 * k <= 256), else 1.  CDX_UNET2_SPLIT=0 switches the mode off, 2 / 4 force a factor (tests)."""
    forced = os.environ.get("CDX_UNET2_SPLIT", "auto")
    if forced == "0" or (forced == "auto" and (os.environ.get("CDX_UNET2_T") or os.environ.get("CDX_UNET2_NW"))):
        return 1                                      # (a forced workgroup shape means the ordinary program)
    for k in (4, 2):
        if forced in ("auto", str(k)) and -(-batch // 8) * 8 * k <= N_CUS:
            return k
    return 1


def group_factor(batch: int) -> int:
    """Trajectories per GROUP of an unconditional U-Net loop at full batch (compile_janner2_group): 4 when the batch needs more than
    128 workgroups but still fits one trajectory per CU (every workgroup of the launch must be resident: ceil(B / (8 k)) * 8 k <= 256),
    else 1.  CDX_UNET2_GROUP=0 switches the mode off, 2 / 4 force a group size (tests, A/B runs)."""
    forced = os.environ.get("CDX_UNET2_GROUP", "auto")
    if forced == "0" or (forced == "auto" and (os.environ.get("CDX_UNET2_T") or os.environ.get("CDX_UNET2_NW"))):
        return 1
    if not (N_CUS // 2 < batch <= N_CUS) and forced == "auto":
        return 1
    for k in (4, 2):
        if forced in ("auto", str(k)) and -(-batch // (8 * k)) * 8 * k <= N_CUS:
            return k
    return 1


_scache = weakref.WeakKeyDictionary()
_gcache2 = weakref.WeakKeyDictionary()


def compiled_group2(module, horizon: int, k: int) -> _Compiled2:
    per = _gcache2.setdefault(module, {})
    sig = R._signature(module)
    key = (horizon, k, P2.group_min_bytes())
    hit = per.get(key)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            comp = _Compiled2(P2.compile_janner2_group(module, horizon, k), sig)
        except ValueError as e:
            comp = _Compiled2(None, sig, str(e))
    per[key] = comp
    return comp



def compiled_split2(module, horizon: int, k: int) -> _Compiled2:
    per = _scache.setdefault(module, {})
    sig = R._signature(module)
    hit = per.get((horizon, k))
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            comp = _Compiled2(P2.compile_janner2_split(module, horizon, k), sig)
        except ValueError as e:
            comp = _Compiled2(None, sig, str(e))
    per[(horizon, k)] = comp
    return comp


_emb_bufs = {}


def cond_film_table(comp: _Compiled2, module, t_vec: torch.Tensor, cond: Optional[torch.Tensor], per_sample: bool = False) -> torch.Tensor:
    """FiLM rows of a CONDITIONAL JannerUNet1d request, one per (timestep, trajectory): the condition embedding enters the time
    embedding before map_emb (reference jannerunet.py:160-164: emb = map_noise(t) + condition), so every block's FiLM vector is
    per-sample.  Row s * B + b; `t_vec` (S,) with a shared condition batch (B, emb_dim), or -- `per_sample`, stand-alone forwards --
    t_vec (B,) = one timestep per sample (S = 1).  One cdx_unet2_embtab launch over S * B rows; two zero spare rows close the table (a half-empty last
    workgroup reads past its batch)."""
    prog = comp.prog
    dev = prog.blob.device
    e = prog.embtabs[0]
    with torch.no_grad():
        temb = R._f32c(module.map_noise(t_vec), dev)
        if per_sample:
            rows = temb if cond is None else temb + cond
        else:
            rows = (temb[:, None, :] + cond[None, :, :]).reshape(-1, temb.shape[1])
        rows = rows.contiguous()
    n = rows.shape[0]
    key = (dev, R._stream_ptr(dev))
    buf = _emb_bufs.get(key)
    if buf is None or buf.numel() < (n + 2) * prog.n_emb:
        _emb_bufs[key] = buf = torch.zeros((n + 2) * prog.n_emb, device=dev, dtype=torch.float32)
    out = buf[: (n + 2) * prog.n_emb].view(n + 2, prog.n_emb)
    args = CdxUnet2EmbtabArgs(wblob=prog.blob.data_ptr(), emb_dim=e["emb_dim"], hidden=e["hidden"], md=e["md"], n_emb=e["n_emb"],
                              w0=e["w0"], b0=e["b0"], w2=e["w2"], b2=e["b2"], w3=e["w3"], b3=e["b3"],
                              temb=rows.data_ptr(), n_rows=n, out=out.data_ptr(), out_ld=prog.n_emb, col0=e["col0"],
                              w4=e["w4"], b4=e["b4"], n_raw=e["n_raw"], col4=e["col4"])
    R._check(_lib().cdx_unet2_embtab(ctypes.byref(args), R._stream_ptr(dev)), "cdx_unet2_embtab")
    return out


def chi_film_table(comp: _Compiled2, module, t_vec: torch.Tensor, cond: Optional[torch.Tensor], per_sample: bool = False,
                   plan=None) -> torch.Tensor:
    """FiLM rows of a ChiUNet1d request (reference chiunet.py:160-163 + :36-45): emb = [map_emb(map_noise(t)) | global_cond_encoder(cond)],
    every block applies Linear(Mish(emb)) -- separable into a per-step and a per-trajectory part, so the table is four GEMMs on
    ``cdx_gemm_f32`` (time MLP with Mish epilogues; condition encoder with Mish; the stacked block Linears once over the encoded
    conditions and once over the (step, trajectory) rows with the condition part added as the GEMM's row-periodic table).
    `cond` (B, cond_dim): rows s * B + b (+ two zero spare rows); `cond` None: the zero-condition table of a classifier-free-guidance
    pair, one row per step (reference diffusionsde.py:175-206 feeds zeros).  `per_sample`: t_vec (B,), one row per sample.  `plan`:
    the solver's cached plan `t_vec` came from (memoises the time half)."""
    from . import blocks
    prog = comp.prog
    dev = prog.blob.device
    f = prog.meta["chi_film"]
    with torch.no_grad():
        # the time half depends on (schedule, weights) only: kept with the solver's cached plan like the JannerUNet1d tables
        memo = plan.__dict__.setdefault("_memo", {}) if plan is not None else None
        key_t = ("chi_te", str(dev), id(module))
        hit = memo.get(key_t) if memo is not None else None
        if hit is None or hit[0] != comp.sig:
            temb = R._f32c(module.map_noise(t_vec), dev)
            h = blocks.linear(temb, module.map_emb[0].weight, module.map_emb[0].bias, act="mish")
            hit = (comp.sig, blocks.linear(h, module.map_emb[2].weight, module.map_emb[2].bias, act="mish"))   # Mish(map_emb(temb)), (S, E)
            if memo is not None:
                memo[key_t] = hit
        te = hit[1]
        gce = module.global_cond_encoder
        c_in = cond if cond is not None else torch.zeros((1, gce.in_features), device=dev, dtype=torch.float32)
        ce = blocks.linear(R._f32c(c_in, dev), gce.weight, gce.bias, act="mish")                     # Mish(encoder(cond)), (B, E)
        cpart = blocks.linear(ce, f["w_c"], f["bias"])                                                # (B, n_emb), block biases included
        n_b = cpart.shape[0]
        if per_sample or cond is None:
            rows_t, n = te, te.shape[0]
            assert per_sample is False or n == n_b
        else:
            rows_t = te.repeat_interleave(n_b, dim=0)                                                 # row s * B + b reads step s
            n = rows_t.shape[0]
        key = (dev, R._stream_ptr(dev), "u" if cond is None else "c")
        buf = _emb_bufs.get(key)
        if buf is None or buf.numel() < (n + 2) * prog.n_emb:
            _emb_bufs[key] = buf = torch.zeros((n + 2) * prog.n_emb, device=dev, dtype=torch.float32)
        out = buf[: (n + 2) * prog.n_emb].view(n + 2, prog.n_emb)
        blocks.linear(rows_t, f["w_t"], None, out=out[:n], table=cpart)                              # + cpart[row % B]
    return out


def fused_sample2(solver, net, plan, xt, prior, feed, fix_mask, x_min, x_max, x_scale: Optional[float] = None,
                  cond=None, w_cfg: float = 0.0) -> Optional[torch.Tensor]:
    """JannerUNet1d, every step kind: the whole loop in one cdx_unet2_run launch.  None -> the caller falls back.
    `x_scale` given: `xt` is the raw N(0, I) draw and the kernel forms x_T = xt * x_scale blended with the prior itself.
    `cond` (B, emb_dim) with w_cfg = 1: conditional forwards (per-trajectory FiLM rows); w_cfg not in {0, 1}: the classifier-free
    guidance pair inside the launch (reference diffusionsde.py:175-206).  EDM / consistency plans (kinds 5-7) included."""
    b, h, d = xt.shape
    if b == 0:
        return torch.empty_like(R._f32c(xt, xt.device))      # empty request: nothing to launch
    if supported(net, h) is not None:
        return None
    edm = R.plan_is_edm(plan)
    use_cond = cond is not None and w_cfg != 0.0
    chi = R._is_chiunet(net)
    if chi and not use_cond:
        return None                                   # ChiUNet1d cannot run unconditionally (the reference raises)
    if (use_cond or edm) and x_scale is not None:
        return None
    dev = xt.device
    comp, parts = plan_for(net, h, b)
    split, plain, group = 0, None, False
    # (split / grouped launches carry per-launch sequence numbers and ticket bases as kernel arguments: a captured launch would replay
    #  stale ones -- under stream capture the ordinary program serves the request)
    capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if not chi and not use_cond and not edm and not comp.prog.compact and not capturing and \
            (R._prof["buf"] is None or os.environ.get("CDX_UNET2_GROUP_PROF") == "1") and _modes_allowed(dev):
        k = split_factor(b) if _split_ok.get(dev, True) and R._prof["buf"] is None else 1      # small batches: one trajectory over k workgroups of an XCD
        if k > 1:
            alt = compiled_split2(net, h, k)
            if alt.prog is not None:
                plain, (comp, parts, split) = (comp, parts), (alt, None, k)
        elif _group_ok.get(dev, True):
            k = group_factor(b)                       # one trajectory per CU: k trajectories over the k workgroups of a group
            if k > 1:
                alt = compiled_group2(net, h, k)
                if alt.prog is not None:
                    plain, (comp, parts, split, group) = (comp, parts), (alt, None, k, True)
    if chi:
        cond = torch.flatten(cond, 1)
        if cond.shape != (b, comp.prog.meta["cond_dim"]):
            return None
    elif use_cond and (cond.dim() != 2 or cond.shape != (b, comp.prog.emb_dim)):
        return None
    with torch.no_grad():
        steps_dev = R.steps_to_device(plan, dev)
        emb_u = None
        if chi:
            emb = chi_film_table(comp, net, R.device_times(plan, dev), R._f32c(cond, dev), plan=plan)
            if w_cfg != 1.0:
                emb_u = chi_film_table(comp, net, R.device_times(plan, dev), None, plan=plan)      # zero condition, one row per step
        elif use_cond:
            emb = cond_film_table(comp, net, R.device_times(plan, dev), R._f32c(cond, dev))
            if w_cfg != 1.0:
                emb_u = plan_film_table(comp, net, plan, dev)                 # zero condition: the per-step table
        else:
            emb = plan_film_table(comp, net, plan, dev)
        noise = feed.many(xt, plan.n_noise)
        xin = R._f32c(xt, dev)
        out = torch.empty_like(xin)
        kw = dict(batch=b, x_in=xin, emb=emb, steps_dev=steps_dev, n_steps=len(plan.steps), predict_noise=R._predicts_noise(plan, solver),
                  prior=R._f32c(prior, dev) if fix_mask is not None else None, fix_mask=fix_mask, noise=noise, x_min=x_min, x_max=x_max,
                  x_scale=x_scale, emb_per_traj=use_cond, emb_u=emb_u, cfg_w=w_cfg, edm=edm)
        launch(comp, x_out=out, parts=parts, split=split, group=group, t_per_wg=1 if split else None, **kw)
        if split and _repair_on():
            # REPAIR launch: the same request on the ordinary program, gated on the error word of the launch above (cdx.h: run_if).  Stream
            # order puts it between that launch and every consumer of `out`: a lost granule (bounded polls -> NaN trajectories) is
            # recomputed before anybody can observe it; when nothing failed every workgroup returns at once (~10 us of a 3.6 ms call).
            launch(plain[0], x_out=out, parts=plain[1], run_if=_split_err(dev), **kw)
        ok = _group_ok if group else _split_ok
        if split and dev not in ok:
            # First split / grouped launch on this device: checked ONCE against the ordinary program on this very request -- a dispatcher
            # that does not give the launch one workgroup per CU shows up as a lost granule or as different numbers, and the mode stays
            # off for the process instead of failing later.
            ref = torch.empty_like(xin)
            launch(plain[0], x_out=ref, parts=plain[1], **kw)
            try:
                check_split_errors(dev, wait=True)
                good = bool(torch.allclose(out, ref, rtol=1e-3, atol=1e-3))
            except RuntimeError as e:
                warnings.warn(f"first-use check of the {'grouped' if group else 'small-batch'} mode: {e}")
                good = False
            ok[dev] = good
            if not good:
                return ref
        elif split and _sync_check(group):
            # CDX_UNET2_SPLIT_SYNC=1: look at the error word of THIS launch before the call returns (the mode then switches off one call
            # earlier).  Not needed for safety any more -- the repair launch has already replaced a failed result.
            torch.cuda.current_stream(dev).synchronize()
            note_exchange_failure(dev)
    return out


def _sync_check(group: bool) -> bool:
    return os.environ.get("CDX_UNET2_SPLIT_SYNC") == "1"


def _repair_on() -> bool:
    return os.environ.get("CDX_UNET2_REPAIR", "1") != "0"       # (=0: A/B runs that price the repair launch; a failed exchange then hands out NaN)


def _modes_allowed(dev) -> bool:
    """Split / grouped launches on `dev`: a whole MI355X only (whole_chip), and not after a lost granule (looked up without a sync)."""
    if dev.type != "cuda":
        return False
    note_exchange_failure(dev)
    return whole_chip(dev)


_split_ok = {}       # device -> did the small-batch mode pass its one-time check there (absent: not checked yet)
_group_ok = {}       # ... the full-batch grouped mode


def stream_bytes_per_forward(prog: P2.Program2, member: int = 0) -> int:
    """Bytes of packed weight records ONE workgroup streams from L2 per denoiser forward: the sum of its work items' record counts
    (1 KiB each).  An ordinary program: the whole packed set; member views of split / grouped programs: 1/k of the ops they cut."""
    ops = prog.meta["member_ops"][member] if "member_ops" in prog.meta else prog.ops
    return 1024 * sum(int(P2.op_item(prog.ops_buffer, op, j)[P2.I2_NQ]) for op in ops for j in range(int(op[P2.W2_NITEMS])))


def route_info(net, horizon: int, batch: int, device) -> dict:
    """How an unconditional JannerUNet1d sampling loop of `batch` trajectories is launched on `device` right now -- what fused_sample2
    decides, for reporting (bench.py's `roofline`): mode "grouped" (k trajectories over the k workgroups of a group), "split" (one
    trajectory over k workgroups) or "plain"; the program, the launch parts, the number of workgroups and the weight bytes a workgroup
    streams per forward."""
    comp, parts = plan_for(net, horizon, batch)
    mode, k = "plain", 1
    if not comp.prog.compact and R._prof["buf"] is None:
        ks = split_factor(batch) if _split_ok.get(device, True) else 1
        if ks > 1 and compiled_split2(net, horizon, ks).prog is not None:
            mode, k, comp, parts = "split", ks, compiled_split2(net, horizon, ks), None
        elif ks == 1 and _group_ok.get(device, True):
            kg = group_factor(batch)
            if kg > 1 and compiled_group2(net, horizon, kg).prog is not None:
                mode, k, comp, parts = "grouped", kg, compiled_group2(net, horizon, kg), None
    if mode in ("split", "grouped"):
        n_wg = N_CUS                      # always one workgroup per CU (the groups are formed from per-XCD tickets)
    else:
        n_wg = sum(-(-cnt // t) for _, cnt, t in parts)
    return {"mode": mode, "k": k, "comp": comp, "parts": parts, "workgroups": n_wg,
            "stream_bytes_per_workgroup_forward": stream_bytes_per_forward(comp.prog)}


def backbone_forward2(module, x, noise_t, condition=None) -> Optional[torch.Tensor]:
    """``JannerUNet1d.forward`` / ``ChiUNet1d.forward`` (per-sample timesteps, condition) in one launch: one FiLM row per sample."""
    b, h, d = x.shape
    if supported(module, h) is not None:
        return None
    if b == 0:
        return torch.empty_like(R._f32c(x, x.device))
    comp, t_wg = shape_for(module, h, b)
    if comp.prog.compact:
        return None                                   # (compact-only nets: stand-alone forwards stay with the implicit-GEMM executor)
    chi = R._is_chiunet(module)
    if chi:
        if condition is None:
            return None                               # the reference raises on a missing condition (Q12): keep that path
        condition = torch.flatten(condition, 1)
        if tuple(condition.shape) != (b, comp.prog.meta["cond_dim"]):
            return None
    elif condition is not None and (condition.dim() != 2 or tuple(condition.shape) != (b, comp.prog.emb_dim)):
        return None
    with torch.no_grad():
        t = noise_t.reshape(-1)
        if t.shape[0] == 1:
            t = t.expand(b)
        if chi:
            emb = chi_film_table(comp, module, t.contiguous(), R._f32c(condition, x.device), per_sample=True)
        else:
            emb = cond_film_table(comp, module, t.contiguous(), None if condition is None else R._f32c(condition, x.device), per_sample=True)
        xin = R._f32c(x, x.device)
        out = torch.empty_like(xin)
        launch(comp, batch=b, x_in=xin, x_out=out, emb=emb, t_per_wg=t_wg, emb_per_traj=True)
    return out


# ------------------------------------------------------------------------------------------------------------------- #
# classifier-guided sampling: denoiser forward + classifier forward/backward in the same launch                          #
# ------------------------------------------------------------------------------------------------------------------- #
_gcache = weakref.WeakKeyDictionary()


def compiled_guided2(net, clf_net, horizon: int, two: bool = False, three: bool = False) -> _Compiled2:
    """Guided program (denoiser ops, then the HalfJannerUNet1d classifier's forward and backward-data ops), 8-wave shape.  `two`: the
    variant for two trajectories per workgroup -- saved tensors in a global workspace, small staging area.  ``.prog is None`` +
    ``.why`` when it does not exist (LDS plan, unsupported layers)."""
    per = _gcache.setdefault(net, {})
    sig = (R._signature(net), R._signature(clf_net))
    key = (id(clf_net), horizon, bool(two), bool(three))
    hit = per.get(key)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            kw = dict(save_global=True, max_stage=GUIDED_T2_STAGE, max_lds_bytes=80 * 1024) if two else {}
            if three:        # compact on top: state / multistep memory in global memory, in-place residual outputs (config 2: 49.6 KB)
                kw = dict(save_global=True, compact=True, max_stage=GUIDED_T2_STAGE, max_lds_bytes=(160 * 1024) // 3 // 16 * 16)
            try:
                comp = _Compiled2(P2.compile_guided2(net, clf_net, horizon, **kw), sig)
            except ValueError:                           # (LDS plan too large)
                if two or three:
                    raise
                # wider nets, one trajectory per workgroup: saved tensors in the global workspace (model_dim 64 at H = 32, the kitchen
                # Diffuser: 141 KB), then also the state / multistep memory in global memory and in-place residual outputs (model_dim
                # 64 at H = 64, the antmaze Diffuser: 158 KB)
                try:
                    comp = _Compiled2(P2.compile_guided2(net, clf_net, horizon, save_global=True), sig)
                except ValueError:
                    comp = _Compiled2(P2.compile_guided2(net, clf_net, horizon, save_global=True, compact=True), sig)
        except ValueError as e:
            comp = _Compiled2(None, sig, str(e))
    per[key] = comp
    return comp


_ggcache = weakref.WeakKeyDictionary()
_gguided_ok = {}     # device -> did the grouped GUIDED mode pass its one-time check there (absent: not checked yet)
_sguided_ok = {}     # ... the small-batch (split) GUIDED mode


def compiled_guided_group2(net, clf_net, horizon: int, k: int) -> _Compiled2:
    """The GROUPED guided program (P2.compile_guided2_group): the denoiser's stream-bound layers computed per member for 1/k of the output
    channels of the group's k trajectories, everything else -- the classifier's forward / backward ops included -- on the member's own
    trajectory.  ``.prog is None`` + ``.why`` when it does not exist."""
    return _compiled_guided_members(net, clf_net, horizon, k, True)


def compiled_guided_split2(net, clf_net, horizon: int, k: int) -> _Compiled2:
    """The SMALL-BATCH guided program (P2.compile_guided2_split): one trajectory over k workgroups of an XCD, the denoiser's ops cut by
    row tiles where that pays, the classifier's ops computed by every member."""
    return _compiled_guided_members(net, clf_net, horizon, k, False)


def _compiled_guided_members(net, clf_net, horizon: int, k: int, group: bool) -> _Compiled2:
    per = _ggcache.setdefault(net, {})
    sig = (R._signature(net), R._signature(clf_net))
    key = (id(clf_net), horizon, k, group, P2.group_min_bytes())
    hit = per.get(key)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            comp = _Compiled2((P2.compile_guided2_group if group else P2.compile_guided2_split)(net, clf_net, horizon, k), sig)
        except ValueError as e:
            comp = _Compiled2(None, sig, str(e))
    per[key] = comp
    return comp


GUIDED_T2_STAGE = 2304       # floats of staging area a two-trajectory guided program may use per op (config 2: 2 x 80.8 KB of LDS)


def guided_supported(net, clf_net, horizon: int) -> Optional[str]:
    if not enabled() or os.environ.get("CDX_UNET2_GUIDED", "1") == "0":
        return "disabled by CDX_UNET2=0 / CDX_UNET2_GUIDED=0"
    why = _structural(net, horizon)          # (not `supported`: a net whose unguided default program does not fit may still have a
    if why is not None:                      #  guided one -- the compact variant)
        return why
    return compiled_guided2(net, clf_net, horizon).why


def guided_sample2(solver, net, clf_net, plan, xt, prior, feed, fix_mask, x_min, x_max, w_cg) -> Optional[torch.Tensor]:
    """Whole classifier-guided loop in ONE cdx_unet2_run launch (reference diffusionsde.py:526-594 with w_cg != 0): per step the
    denoiser forward, the classifier's forward + backward-data pass and the shifted, clipped solver update, all on LDS-resident
    state.  None -> the caller takes the per-step executor (cdx_guided_run)."""
    b, h, d = xt.shape
    if R.plan_is_edm(plan) or any(st.kind > 2 for st in plan.steps) or guided_supported(net, clf_net, h) is not None:
        return None
    dev = xt.device
    comp, t = compiled_guided2(net, clf_net, h), 1
    forced = os.environ.get("CDX_UNET2_T")
    parts = None
    if forced is None and b > 2 * N_CUS and os.environ.get("CDX_UNET2_T3", "1") != "0":
        # rounds of 256 x 3 trajectories on the compact variant, the remainder at whatever finishes it soonest (same program)
        alt = compiled_guided2(net, clf_net, h, three=True)
        if alt.prog is not None:
            cut = plan_parts(b, 3, GUIDED_ROUND_COST)
            if max(p[2] for p in cut) == 3:
                comp, t, parts = alt, 3, cut
    if forced == "3":
        alt = compiled_guided2(net, clf_net, h, three=True)
        if alt.prog is not None:
            comp, t = alt, 3
    if parts is None and (forced == "2" or (forced is None and b > 256)):
        alt = compiled_guided2(net, clf_net, h, two=True)
        if alt.prog is not None:
            comp, t = alt, 2
    from .plan import cached
    pn = R._predicts_noise(plan, solver)
    with torch.no_grad():
        with_logp = os.environ.get("CDX_UNET2_LOGP", "1") != "0"       # (A/B hook: 0 = the classifier's own log_p launch afterwards)
        emb = plan_film_table(comp, net, plan, dev, modules=[net, clf_net], zero_row=with_logp)
        steps_dev = R.steps_to_device(plan, dev)
        cg = cached(plan, ("cg2", str(dev), float(w_cg), int(pn)), lambda: torch.tensor(
            [(-(w_cg * st.sigma)) if pn else (w_cg * ((st.sigma ** 2) / st.alpha)) for st in plan.steps], dtype=torch.float32, device=dev))
        noise = feed.many(xt, plan.n_noise)
        xin = R._f32c(xt, dev)
        out = torch.empty_like(xin)
        logp = torch.empty((b, 1), dtype=torch.float32, device=dev) if with_logp else None
        kw = dict(batch=b, x_in=xin, emb=emb, steps_dev=steps_dev, n_steps=len(plan.steps), predict_noise=pn,
                  prior=R._f32c(prior, dev) if fix_mask is not None else None, fix_mask=fix_mask, noise=noise, x_min=x_min,
                  x_max=x_max, cg_scale=cg)
        # GROUPED guided launch (round 6): one trajectory per CU, groups of k workgroups share the weight stream of the denoiser's
        # stream-bound layers -- the unguided grouped mode (`sample2`) with the classifier's ops on each member's own trajectory; same
        # gates (whole chip, no capture, no lost granule so far), same repair launch, same one-time check against the ordinary program
        # ... and below 129 trajectories the SMALL-BATCH form: one trajectory over 2 or 4 workgroups, the denoiser's ops cut by row tiles
        # (`compile_guided2_split`), the classifier's ops computed by every member (CDX_UNET2_GUIDED_SPLIT=0: off)
        gk, galt, ggroup = 1, None, True
        capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if forced is None and t == 1 and parts is None and with_logp and not capturing and R._prof["buf"] is None and _modes_allowed(dev):
            if os.environ.get("CDX_UNET2_GUIDED_GROUP", "1") != "0" and _gguided_ok.get(dev, True):
                gk = group_factor(b)
                if gk > 1:
                    galt = compiled_guided_group2(net, clf_net, h, gk)
            if galt is None and os.environ.get("CDX_UNET2_GUIDED_SPLIT", "1") != "0" and _sguided_ok.get(dev, True):
                gk = split_factor(b)
                if gk > 1:
                    galt, ggroup = compiled_guided_split2(net, clf_net, h, gk), False
            if galt is not None and galt.prog is None:
                gk, galt = 1, None
        if galt is not None:
            ok_map = _gguided_ok if ggroup else _sguided_ok
            gemb = plan_film_table(galt, net, plan, dev, modules=[net, clf_net], zero_row=with_logp)
            launch(galt, x_out=out, t_per_wg=1, split=gk, group=ggroup, logp_out=logp, **{**kw, "emb": gemb})
            if _repair_on():
                launch(comp, x_out=out, t_per_wg=t, logp_out=logp, run_if=_split_err(dev), **kw)
            if dev not in ok_map:
                ref, ref_logp = torch.empty_like(xin), torch.empty_like(logp)
                launch(comp, x_out=ref, t_per_wg=t, logp_out=ref_logp, **kw)
                try:
                    check_split_errors(dev, wait=True)
                    good = bool(torch.allclose(out, ref, rtol=1e-3, atol=1e-3)) and bool(torch.allclose(logp, ref_logp, rtol=1e-3, atol=1e-3))
                except RuntimeError as e:
                    warnings.warn(f"first-use check of the {'grouped' if ggroup else 'small-batch'} guided mode: {e}")
                    good = False
                ok_map[dev] = good
                if not good:
                    out, logp = ref, ref_logp
            elif _sync_check(True):
                torch.cuda.current_stream(dev).synchronize()
                note_exchange_failure(dev)
        else:
            launch(comp, x_out=out, t_per_wg=t, parts=parts, logp_out=logp, **kw)
        if logp is not None:
            # the classifier's score of the FINAL trajectories (timestep 0) came out of the same launch: handed to the solver's
            # post-processing on the tensor itself (diffusionsde._sample_common reads it instead of calling classifier.logp)
            out._cdx_logp = logp
    return out


def classifier_gradient2(net, clf_net, x, noise_t) -> Optional[torch.Tensor]:
    """d classifier(x, t).sum() / d x through the guided program in forward mode (ONE timestep for the batch) -- test / probe entry."""
    b, h, d = x.shape
    if guided_supported(net, clf_net, h) is not None:
        return None
    comp = compiled_guided2(net, clf_net, h)
    with torch.no_grad():
        emb = film_table(comp, net, noise_t.reshape(-1)[:1], modules=[net, clf_net])
        xin = R._f32c(x, x.device)
        out = torch.empty_like(xin)
        launch(comp, batch=b, x_in=xin, x_out=out, emb=emb, t_per_wg=1, with_backward=True)
    return out


_ccache = weakref.WeakKeyDictionary()


def compiled_classifier2(clf_net, horizon: int) -> _Compiled2:
    """The classifier's own program (forward ops + head, engine/program2.py:compile_classifier2); ``.prog is None`` + ``.why`` when
    the v2 compiler does not take it."""
    per = _ccache.setdefault(clf_net, {})
    sig = R._signature(clf_net)
    hit = per.get(horizon)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            comp = _Compiled2(P2.compile_classifier2(clf_net, horizon), sig)
        except ValueError as e:
            comp = _Compiled2(None, sig, str(e))
    per[horizon] = comp
    return comp


def classifier_forward2(clf_net, x, noise_t) -> Optional[torch.Tensor]:
    """``HalfJannerUNet1d.forward`` with out_dim 1 -- what ``CumRewClassifier.logp`` evaluates (reference classifier/base.py:62-72,
    nn_classifier/half_jannerunet.py:102-125) -- as the log_p pass of ONE cdx_unet2_run launch: per-sample timesteps through one
    FiLM row per trajectory.  (batch, 1), or None when the v2 compiler does not take the classifier."""
    if not enabled() or x.dim() != 3 or os.environ.get("CDX_UNET2_CLASSIFIER", "1") == "0":
        return None
    b, h, d = x.shape
    if getattr(clf_net, "in_dim", None) != d:
        return None
    if b == 0:
        return torch.empty((0, 1), dtype=torch.float32, device=x.device)
    comp = compiled_classifier2(clf_net, h)
    if comp.prog is None:
        return None
    with torch.no_grad():
        t = noise_t.reshape(-1)
        if t.shape[0] == 1:
            t = t.expand(b)
        emb = film_table(comp, clf_net, t.contiguous())
        xin = R._f32c(x, x.device)
        out = torch.empty((b, 1), dtype=torch.float32, device=x.device)
        launch(comp, batch=b, x_in=xin, x_out=xin, emb=emb, t_per_wg=1, with_backward=True, logp_out=out)
    return out


# ------------------------------------------------------------------------------------------------------------------- #
# batch-tiled MLP denoisers (PearceMlp, DQLMlp / DVInvMlp, MlpNNDiffusion, SfBCUNet) on the second-generation kernel     #
# ------------------------------------------------------------------------------------------------------------------- #
_mcache = weakref.WeakKeyDictionary()


def compiled_mlp2(net, kind: str, tile: int) -> _Compiled2:
    per = _mcache.setdefault(net, {})
    sig = R._signature(net)
    hit = per.get(tile)
    if hit is not None and hit.sig == sig:
        return hit
    with torch.no_grad():
        try:
            comp = _Compiled2(P2.MLP2_COMPILERS[kind](net, tile), sig)
        except ValueError as e:                        # widths the epilogue partition / GroupNorm layout does not take (the
            comp = _Compiled2(None, sig, str(e))        #  documented signal; invariant failures of the compiler propagate)
    per[tile] = comp
    return comp


def mlp_table(comp: _Compiled2, net, plan, device) -> torch.Tensor:
    """(steps, n_emb) table of a batch-tiled MLP program: bias rows = static bias + the layer's time-dependent inputs (time embedding,
    the net's own time MLP over it -- on ``cdx_gemm_f32`` --, the raw timestep column).  Kept with the solver's cached plan."""
    from . import blocks
    memo = plan.__dict__.setdefault("_memo", {})
    key = ("mlp2", str(device), id(net), comp.prog.meta["mlp"]["tile"])
    hit = memo.get(key)
    if hit is not None and hit[0] == comp.sig:
        return hit[1]
    spec = comp.prog.meta["mlp"]
    r = spec["rows"]
    with torch.no_grad():
        t_vec = R.device_times(plan, device)
        temb = R._f32c(net.map_noise(t_vec), device)
        out = r["bias"][None, :].repeat(temb.shape[0], 1).contiguous()
        if "temb" in r:
            out = blocks.linear(temb, r["temb"], None, residual=out)
        if "tfeat" in r:
            seq = net.time_mlp if spec["kind"] == "dql" else net.t_layer
            act = "mish" if spec["kind"] == "dql" else "silu"
            tf = blocks.linear(blocks.linear(temb, seq[0].weight, seq[0].bias, act=act), seq[2].weight, seq[2].bias)
            out = blocks.linear(tf, r["tfeat"], None, residual=out)
        if "t" in r:                                    # the raw timestep as a feature (PearceMlp, Q11): a rank-1 term
            out = out + t_vec.to(torch.float32)[:, None] * r["t"][:, 0][None, :]
    memo[key] = (comp.sig, out.contiguous())
    return memo[key][1]


def fused_sample_mlp2(solver, net, kind, plan, xt, prior, cond_vec, w_cfg, feed) -> Optional[torch.Tensor]:
    """Batch-tiled MLP denoisers (x of shape (B, D)): one workgroup per `tile` samples, the whole loop in ONE cdx_unet2_run launch
    (conditional forward, the classifier-free-guidance pair and EDM plans included).  None -> the caller falls back."""
    if not enabled() or os.environ.get("CDX_UNET2_MLP", "1") == "0":
        return None
    b, d = xt.shape
    dev = xt.device
    if b == 0:
        return torch.empty_like(R._f32c(xt, dev))
    if cond_vec is None and w_cfg not in (0.0, 1.0):
        return None                                   # the reference raises here; let the torch executor do it
    use_cond = cond_vec is not None and w_cfg != 0.0
    if kind == "dql" and type(net).__name__ == "DVInvMlp" and not use_cond:
        return None                                   # DVInvMlp requires its condition (the reference fails on None)
    try:
        fix_mask = R._dense_hd(solver.fix_mask, 1, d, dev)
        clip = getattr(plan, "clip_each_step", True)
        x_min = R._dense_hd(getattr(solver, "x_min", None), 1, d, dev) if clip else None
        x_max = R._dense_hd(getattr(solver, "x_max", None), 1, d, dev) if clip else None
    except (ValueError, RuntimeError):
        return None
    tile, comp = R.mlp_tile(b), None
    while tile >= 4:                                  # wide nets: fewer samples per workgroup keep the epilogue partition in range
        comp = compiled_mlp2(net, kind, tile)
        if comp.prog is not None:
            break
        tile //= 2
    if comp is None or comp.prog is None:
        return None
    n_tiles = -(-b // tile)
    pad = n_tiles * tile - b

    def rows(t):                                      # (B, D) -> (n_tiles * tile, D), zero rows appended
        t = R._f32c(t, dev)
        return torch.cat([t, t.new_zeros(pad, *t.shape[1:])]) if pad else t

    def table(t):                                     # (1, D) -> (tile, D): the kernel indexes bounds / masks per tile row
        return None if t is None else t.expand(tile, d).contiguous()

    ctx = None
    if use_cond:
        ctx = rows(torch.flatten(cond_vec, 1))
        if ctx.shape[1] != comp.prog.meta["mlp"]["cond_dim"]:
            return None
    edm = R.plan_is_edm(plan)
    with torch.no_grad():
        emb = mlp_table(comp, net, plan, dev)
        steps_dev = R.steps_to_device(plan, dev)
        noise = feed.many(xt, plan.n_noise)
        if noise is not None and pad:                 # zero rows behind every draw: the last tile's unused samples
            noise = torch.cat([R._f32c(noise, dev), noise.new_zeros(noise.shape[0], pad, d)], dim=1).contiguous()
        xin = rows(xt)
        out = torch.empty_like(xin)
        pair = use_cond and w_cfg != 1.0
        launch(comp, batch=n_tiles, x_in=xin, x_out=out, emb=emb, steps_dev=steps_dev, n_steps=len(plan.steps),
               predict_noise=R._predicts_noise(plan, solver), prior=rows(prior) if fix_mask is not None else None,
               fix_mask=table(fix_mask), noise=noise, x_min=table(x_min), x_max=table(x_max), t_per_wg=1,
               emb_u=emb if pair else None, cfg_w=w_cfg, edm=edm, ctx=ctx)
    return out[:b]


def describe_forward(comp: _Compiled2, *, batch: int, emb: torch.Tensor, t_per_wg: int, emb_per_traj: bool) -> CdxUnet2Launch:
    """A forward-mode launch description (n_steps = 0) for a caller that fills in x_in / x_out / the step's emb rows itself: the
    denoiser slot of ``cdx_guided_run`` (engine/guided.py).  The caller keeps `comp` and `emb` alive."""
    prog = comp.prog
    if prog.compact or "mlp" in prog.meta:
        raise ValueError("forward-mode descriptions exist for U-Net programs with the state in LDS")
    return CdxUnet2Launch(
        ops=comp.ops_dev.data_ptr(), wblob=prog.blob.data_ptr(), n_ops=len(prog.ops), traj_floats=prog.traj_floats, traj_per_wg=t_per_wg,
        n_waves=prog.nw, tune=int(os.environ.get("CDX_UNET2_TUNE", DEFAULT_TUNE)), x_off=prog.x_off, x_stride=prog.x_stride,
        pred_off=prog.pred_off, pred_stride=prog.pred_stride, prev_off=prog.prev_off, stage_off=prog.stage_off, batch=batch,
        horizon=prog.horizon, dim=prog.dim, traj_first=0, traj_count=batch, emb=emb.data_ptr(), emb_ld=emb.shape[1], n_steps=0,
        x_scale=1.0, grad_off=prog.grad_off, grad_stride=prog.grad_stride, ws_floats=0, emb_per_traj=int(emb_per_traj), n_pass=1, cfg_w=1.0)
