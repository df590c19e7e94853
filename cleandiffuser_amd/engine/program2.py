"""Network -> device program for the fused program kernel (``csrc/cdx_unet2.hip``): one launch = the whole ``sample()`` loop,
activations resident in LDS, weights streamed as 1-KiB MFMA records.  (The round-1 compiler ``program.py`` and its kernel were deleted
in round 3; this is the only program format.)  The format is shaped by what the round-2 profiles showed: with one workgroup per CU the
kernel is bound by per-op latency and by *instruction issue* (a wave64 issues about one instruction every four clocks), so everything the
K loop does per weight record beyond "wait, 4 MFMAs, one LDS read, one global load" was designed out on the host side:

* **4 or 8 wave64 per workgroup** (one or two per SIMD; a program is compiled for ONE shape, ``Program2.nw``); a conv's output
  is cut into (row tile x column group) *tiles*, one work item per wave; a layer with fewer tiles than waves splits K over the
  spare waves.  Items are 8-word records; a wave's first item sits INSIDE the op descriptor (words ``W2_ITEM0 + 8 * wave``...),
  so it is found at a fixed offset without a dependent load.
* **A work item reads ONE source slot with a row that is LINEAR in the tap.**  Slots carry a ``HALO2``-row zero halo, so zero
  padding needs no predicate and a tap change is one per-lane add.  A channel concat (reference jannerunet.py:193
  ``torch.cat([x, skip])``) is never materialised and never switched inside the loop: the K slices of such a layer are cut
  at the boundary between the two sources (the layers that concatenate have at most two tiles, so K is split anyway) and the
  item record names its source.  ``ConvTranspose1d(4, 2, 1)`` is lowered to two 2-tap convs, one per output parity (even
  outputs use taps 3,1 on rows m-1,m; odd outputs taps 2,0 on rows m,m+1), each with its own record stream -- no half-empty
  MFMAs and no parity test in the loop.
* **The 1x1 skip conv of a ResidualBlock** (jannerunet.py:58, :69) is a plain op whose epilogue adds into the block's output.
* **The per-block FiLM vectors** ``Linear(Mish(map_emb(temb)))`` depend only on the step, not on the trajectory: they are
  evaluated once per (weights version, schedule) into a ``(steps, n_emb)`` table by ``cdx_unet2_embtab`` and read as
  per-channel float4 epilogue parameters; the embedding MLP ops disappear from the per-forward program.
* **Epilogue**: staged partial tiles -> 32 lanes per GroupNorm group, float4 of consecutive channels per lane, single-pass
  shifted statistics in registers, Mish, + FiLM vector, + residual, float4 store; the producing op also rewrites the four
  halo rows of its destination (the arena recycles LDS between slots of different shapes).
* **T trajectories per workgroup** (1 or 2): each trajectory owns an identical LDS region (``traj_floats`` apart), so all
  offsets below are relative to the trajectory base.

Descriptor words ``W2_*`` / item words ``I2_*`` MUST mirror ``csrc/cdx_ops2.h`` (tests/test_abi_contract.py checks).
Activations: channel-last rows ``slot[(pos + HALO2) * stride + c]``, ``stride = pad16(C) + 4``.  Conv records: ``_records`` below
(MODE_16X16: lane = k4 * 16 + row, 16 K values per record; MODE_4X4: lane = row, 4 K values).

Round 4 added GROUPED programs (``compile_janner2_group``): k trajectories over the k workgroups of a group, the stream-bound layers
computed per member for 1/k of the output channels of all k trajectories -- see that function.
"""
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .consts import (GN_EPS, MODE_16X16, MODE_4X4, _conv1d_eff, _convT1d_eff, _fbits, slot_stride, supports_janner)

HDR_WORDS = 32                # 25 descriptor words, padded
ITEM2_WORDS = 8
NW2 = 4                       # waves per workgroup, default shape (one per SIMD)
NW2_MAX = 8                   # ... two per SIMD
RING2 = 16                    # weight records in flight per wave, 4-wave shape
RING2_NW8 = 8                 # ... 8-wave shape
GROUPS2 = 8                   # GroupNorm groups the epilogue partition is built for (32 lanes per group)
MAX_NK2 = 4                   # float4 items a lane may own in the epilogue
HALO2 = 2                     # zero rows on either side of a slot: covers kernel sizes <= 5 (pad <= 2)
MIN_SLICE = int(os.environ.get("CDX_UNET2_MIN_SLICE", "3"))      # shortest K slice (records) worth a wave of its own
FUSE_SKIP = os.environ.get("CDX_UNET2_FUSE_SKIP", "1") != "0"    # 1x1 skip convs ride in their block's second conv op ...
FUSE_MAX_RECORDS = int(os.environ.get("CDX_UNET2_FUSE_MAX", "100"))   # ... unless that leaves the main conv K slices longer than this
SPLIT_MIN_RECORDS = int(os.environ.get("CDX_UNET2_SPLIT_MIN", "8"))  # split programs: an op is cut over the members only if a wave of
                                                                       # the uncut op streams at least this many records (an
                                                                       # exchange costs ~3 k cycles, a record ~170 per wave).
                                                                       # Round 6: 32 -> 8 (config 2: 15 -> 21 cut ops; same-box sweep
                                                                       # 0 / 4 / 8 / 12 / 16 / 24 / 32 / 48 / 64: B = 32 3.167 -> 3.133 ms,
                                                                       # profiles/r06_split_min_sweep.txt)

(W2_KIND, W2_FLAGS, W2_COUT, W2_LOUT, W2_LCOLS, W2_CSTRIDE, W2_OSTRIDE, W2_MODE, W2_NT, W2_NITEMS, W2_ITEMS, W2_DST,
 W2_DST_STRIDE, W2_SSTRIDE, W2_KSPLIT, W2_BOFF, W2_GAMMA, W2_BETA, W2_EMB, W2_RES, W2_RES_STRIDE, W2_CG4_SHIFT, W2_INV_CNT,
 W2_NK, W2_COUTP, W2_SAVE, W2_SAVE_STRIDE, W2_STATS, W2_DST2, W2_DST2_STRIDE, W2_KPOST, W2_PBIAS) = range(32)
KIND2_CONV, KIND2_HEAD, KIND2_LOADX, KIND2_LOADC = 0, 1, 2, 3
W2_ITEM0 = 32                 # items 0..nw-1 inline; items nw.. in the tail table at W2_ITEMS


def op_words(nw: int) -> int:
    """int32 words per op descriptor of a program compiled for `nw` waves (cdx.h: CDX2_OP_WORDS(n_waves))."""
    return HDR_WORDS + ITEM2_WORDS * nw


def ring_depth(nw: int) -> int:
    return {4: RING2, 8: RING2_NW8}[nw]
# item record: blob offset of the first record, record count, start cursor (tap | chunk << 8), stage offset of the partial tile,
# first column, (pad | output-position offset << 8), source slot (float offset | row stride << 16), chunks per tap
I2_WOFF, I2_NQ, I2_TAPCC, I2_PART, I2_COL0, I2_PADOOFF, I2_SRCSTR, I2_CCN = range(8)

# F2_SAVE: a GroupNorm op also stores the normalised (pre-affine) values and the group rstd for the backward pass;
# F2_GNBWD: the epilogue is the BACKWARD of (GroupNorm -> Mish) of the layer whose saved values W2_SAVE / W2_STATS name, applied to
# the conv result (+ residual slot); F2_DUAL: the value before that backward is also stored (slot W2_DST2); F2_SAVE_GLOBAL: the saved
# values live in the launch's global workspace (W2_SAVE = float offset inside the trajectory's block) instead of an LDS slot
# F2_FILM (with F2_EMB): the table row holds [scale (pad32(C)) | bias (pad32(C))] at W2_EMB and the epilogue applies scale * y + bias
# (ChiUNet1d's cond_predict_scale FiLM, reference chiunet.py:41-45) instead of y + vector
F2_GN, F2_EMB, F2_RES, F2_PRED, F2_SAVE, F2_GNBWD, F2_DUAL, F2_SAVE_GLOBAL, F2_FILM = 1, 2, 4, 8, 16, 32, 64, 128, 256
# Batch-tiled MLP programs (a "trajectory" = `tile` samples, one per position; a Linear = a 1-tap conv; MLP kernel instantiation only):
# F2_COLNORM (with F2_GN): statistics per POSITION over the group's channels (a per-sample GroupNorm, reference pearcemlp.py FCBlock);
# F2_BIAS_EMB: the bias vector is read from the per-step table row at W2_BOFF (bias + the time-dependent part of the layer's input);
# F2_OUT_DIV: the stored value is divided by the float in W2_ODIV (PearceMlp's h / 1.414); bits 12-15: activation id + 1 of
# consts.ACT_* (0 = the default: Mish after a GroupNorm, none otherwise), applied after the norm, before + emb / + residual
F2_COLNORM, F2_BIAS_EMB, F2_OUT_DIV = 512, 1024, 2048
F2_ACT_SHIFT = 12
W2_ODIV = 26                  # forward ops: alias of W2_SAVE_STRIDE (a backward-pass word)
W2_XG = 28                    # forward ops of a SPLIT program (one trajectory over k workgroups of an XCD): alias of W2_DST2 --
                              # lane groups [lo, hi) this member computes, as lo | hi << 8 | 1 << 16; 0 = the op is not split
# Grouped programs (k trajectories over the k workgroups of a group, one XCD; compile_janner2_group): bits of W2_XG above the lane-group
# range.  XG_XCHG: the members exchange this op's output afterwards; XG_GOP: a GROUPED op -- the member computes its 1/k of the output
# channels for ALL k trajectories (tile columns = trajectory x position, W2_GMAP maps a column to its input row); XG_TRAJ: an ordinary op
# that wrote the member's own trajectory into a group slot -- the exchange gathers whole trajectories instead of channel ranges.
XG_XCHG, XG_GOP, XG_TRAJ = 1 << 16, 1 << 17, 1 << 18
W2_GMAP = 25                  # forward ops of a grouped program: alias of W2_SAVE -- log2(positions per trajectory) | rows per sub-slot << 8
GROUP_MIN_BYTES = None        # an op is grouped only if it streams at least this many weight bytes (an exchange costs ~3.6 k cycles);
                              # None: CDX_UNET2_GROUP_MIN_KB (default 400), read when a program is compiled; tests set a number


def group_min_bytes() -> int:
    return GROUP_MIN_BYTES if GROUP_MIN_BYTES is not None else int(os.environ.get("CDX_UNET2_GROUP_MIN_KB", "400")) * 1024
W2_CGREAL4 = 29               # F2_COLNORM ops: alias of W2_DST2_STRIDE -- float4 items of a lane group that hold REAL channels (0 = all):
                              # a per-sample GroupNorm whose groups are narrower than the lane group keeps zero pad channels out of its variance


def pad32(c: int) -> int:
    """Channel count the epilogue partitions: 32 x a power of two (8 lane groups of 4 x 2^k channels each)."""
    n = (c + 31) // 32
    return 32 * (1 << (n - 1).bit_length())


@dataclass
class Act:
    """An LDS slot of (L + 2 HALO2) channel-last rows of C channels."""
    length: int
    chans: int
    uid: int
    off: int = -1                     # float offset of the slot (its first halo row) inside the trajectory region
    persistent: bool = False
    halo: int = HALO2                 # 0: a slot that is never a conv source (saved normalised values of the backward pass)
    in_global: bool = False           # saved values kept in the per-trajectory global workspace, not in LDS
    gcap: bool = False                # grouped programs: short enough to hold the group's k trajectories side by side (k x length <= 16 columns)
    gk: int = 1                       # ... and, once a grouped op touches it, it does: k sub-slots (each with its own halo rows), `sub_floats` apart

    @property
    def stride(self):
        return slot_stride(self.chans)

    @property
    def sub_floats(self):
        return (self.length + 2 * self.halo) * self.stride

    @property
    def floats(self):
        return self.sub_floats * self.gk

    @property
    def data_off(self):               # float offset of position 0
        return self.off + self.halo * self.stride


@dataclass
class Program2:
    ops: np.ndarray                    # int32 [n_ops, op_words(nw)]
    ops_buffer: np.ndarray             # int32 1-D: ops followed by the item tables
    blob: torch.Tensor                 # float32 1-D on the module's device
    traj_floats: int                   # LDS floats per trajectory
    x_off: int                         # position 0 of the state slot (halo rows lie before it)
    x_stride: int
    pred_off: int
    pred_stride: int
    prev_off: int
    stage_off: int
    horizon: int
    dim: int
    emb_dim: int                       # width of one map_noise(t) row
    n_emb: int                         # width of one FiLM-table row
    embtab: Dict[str, int] = field(default_factory=dict)   # blob offsets / sizes of the embedding MLP
    macs_per_forward: int = 0
    n_conv: int = 0
    meta: dict = field(default_factory=dict)
    nw: int = NW2                      # waves per workgroup the work items were cut for
    embtabs: List[dict] = field(default_factory=list)      # one embedding-MLP spec per network (denoiser[, classifier])
    grad_off: int = -1                 # guided programs: position 0 of the classifier-gradient slot
    grad_stride: int = 0
    ws_floats: int = 0                 # floats of global workspace per trajectory (saved x_hat tensors / compact: multistep memory)
    compact: bool = False              # state and multistep memory in global memory (three trajectories per workgroup)

    def lds_bytes(self, traj_per_wg: int) -> int:
        return 4 * self.traj_floats * traj_per_wg


def _records(w_eff: torch.Tensor, mode: int) -> Tuple[torch.Tensor, int]:
    """w_eff [C_out][taps][C_in] -> records [n_row_tiles][taps * cc][64][4] and cc = chunks per tap.
    Lane layouts: MODE_16X16 lane = k4*16 + row, 16 K per record; MODE_4X4 lane = row, 4 K (asserted on silicon by
    tests/test_gpu_parity.py::test_mfma_lane_maps_on_silicon)."""
    c_out, taps, cs = w_eff.shape
    rows, kch = (16, 16) if mode == MODE_16X16 else (64, 4)
    n_ct = -(-c_out // rows)
    csp = -(-cs // kch) * kch
    wp = torch.zeros(n_ct * rows, taps, csp, device=w_eff.device, dtype=torch.float32)
    wp[:c_out, :, :cs] = w_eff
    cc = csp // kch
    if mode == MODE_16X16:
        wp = wp.reshape(n_ct, 16, taps, cc, 4, 4).permute(0, 2, 3, 4, 1, 5)
    else:
        wp = wp.reshape(n_ct, 64, taps, cc, 4).permute(0, 2, 3, 1, 4)
    return wp.reshape(n_ct, taps * cc, 64, 4).contiguous(), cc


class _Builder2:
    def __init__(self, device, nw: int = NW2):
        if nw not in (NW2, NW2_MAX):
            raise ValueError(f"v2 programs are compiled for {NW2} or {NW2_MAX} waves, not {nw}")
        self.device = device
        self.nw = nw
        self.ops: List[List[int]] = []
        self.op_acts: List[dict] = []                      # per op: srcs / res / dst / save / dst2 slots + reads / writes (liveness)
        self.op_item_src: List[List[int]] = []            # per op, per item: index of the source slot it reads
        self.op_items: List[list] = []
        self.chunks: List[torch.Tensor] = []
        self.blob_len = 0
        self.acts: List[Act] = []
        self.stage = 0
        self.macs = 0
        self.n_emb = 0
        self.n_stats = 0
        self.max_stage = 1 << 30           # cap (floats) on the staging area of one op = K slices x positions x row stride (guided
                                           # programs for two trajectories per workgroup keep it small)
        self.alias_residual = False        # write identity-residual block outputs in place over their input slot
        self.wrap_live: List[Act] = []     # non-persistent slots that the solver step writes for op 0 of the next forward (the state)
        self.keep_to_end: List[Act] = []   # arena slots the solver step reads after the last op (prediction / gradient of a guided program)
        self.save_global = False
        self.ws_floats = 0                 # per-trajectory global workspace (saved x_hat tensors) when save_global
        self.allow_4x4 = True
        self.fuse_max = FUSE_MAX_RECORDS   # longest main-conv K slice (records) next to which an extra conv still rides in the same op
        self.member = (0, 1)               # (m, k): this builder emits member m's view of a program split over k workgroups -- the
                                           # member computes 1/k of the row tiles of every op that can be cut that way (conv())
        self.xchg_floats = 0               # largest tile (positions x padded channels) a split op exchanges
        self.grouped = False               # member views of a GROUPED program (k trajectories over k workgroups): see conv()
        self.group_on = True               # ... ops lowered while this is off are neither grouped nor cut (the classifier's part of a guided program)

    def add(self, t: torch.Tensor, pad_to: int = 4) -> int:
        t = t.detach().to(device=self.device, dtype=torch.float32).reshape(-1)
        off = self.blob_len
        pad = (-t.numel()) % pad_to
        if pad:
            t = torch.cat([t, torch.zeros(pad, device=self.device)])
        self.chunks.append(t)
        self.blob_len += t.numel()
        return off

    def act(self, length: int, chans: int, persistent=False, halo: int = HALO2) -> Act:
        a = Act(length, chans, len(self.acts), persistent=persistent, halo=halo)
        k = self.member[1]
        a.gcap = bool(self.grouped and k > 1 and not persistent and halo == HALO2 and length & (length - 1) == 0 and k * length <= 16)
        self.acts.append(a)
        return a

    def save_slot(self, length: int, chans: int) -> Act:
        """Where a GroupNorm layer keeps its normalised values for the backward pass: an LDS slot without halo, or (save_global) a
        block of the trajectory's global workspace."""
        a = self.act(length, chans, halo=0)
        if self.save_global:
            a.in_global, a.persistent = True, True       # (persistent: not part of the LDS arena)
            a.off = self.ws_floats
            self.ws_floats += (a.floats + 3) // 4 * 4
        return a

    def emb_slot(self, c_out: int, film: bool = False) -> int:
        """Columns of the FiLM table for one block: pad32(C) (additive vector) or, `film`, [scale | bias] = 2 x pad32(C)."""
        off = self.n_emb
        self.n_emb += pad32(c_out) * (2 if film else 1)
        return off

    def stats_slot(self) -> int:
        """8 floats (one rstd per GroupNorm group) in the persistent statistics area; returns the float index."""
        off = self.n_stats
        self.n_stats += GROUPS2
        return off

    def conv(self, srcs: List[Act], dst: Act, w_eff: torch.Tensor, bias: Optional[torch.Tensor], *, stride=1, pad=0,
             transposed=False, phases=None, gn: Optional[nn.Module] = None, emb_off: int = -1, res: Optional[Act] = None,
             pred: bool = False, save: Optional[Tuple[Act, int]] = None, bwd: Optional[dict] = None,
             extra: Optional[List[dict]] = None, film: bool = False, act: Optional[int] = None, col_norm: bool = False,
             bias_row: int = -1, out_div: Optional[float] = None, cg_real: int = 0):
        """One fused op: conv over the channel concat of `srcs` -> epilogue -> dst.

        `extra`: further stride-1 convs with the same output shape computed by the SAME op on other waves -- dicts
        {srcs, w_eff, pad, bias, post}.  post=False: their partial tiles are simply summed with the main conv's (the two convs of a
        backward residual sum).  post=True: their sum (+ bias) is added AFTER the GroupNorm / Mish / FiLM of the main conv -- the
        1x1 skip conv of a ResidualBlock (reference jannerunet.py:58, :69) rides in its second conv's op instead of costing an op.

        Forward epilogue: [GroupNorm -> Mish] -> [+ emb | `film`: scale * y + bias] -> [+ residual slot `res` (may be `dst`: accumulate)].  `save=(slot, stats)`
        additionally stores the normalised pre-affine values and the group rstd (what the backward of this layer needs).
        Backward epilogue (`bwd=dict(gn=<GroupNorm of the layer below>, save=<its saved slot>, stats=<its stats index>,
        dst2=<slot or None>)`): v = conv [+ res]; dst2 <- v; dst <- d/du of Mish(GroupNorm(u)) applied to v.

        `transposed`: ConvTranspose1d(k=4, stride=2, pad=1) as two 2-tap convs, one per output parity.  `phases`: explicit list of
        (taps-in-row-order weights [C_out][taps][C_in], item pad, output offset) of a stride-2 scatter (the backward of a strided
        conv); phases may have different tap counts."""
        c_out, taps, c_in = w_eff.shape
        l_out = dst.length
        assert c_in == sum(a.chans for a in srcs) and dst.chans == c_out and 1 <= len(srcs) <= 2
        gop = False
        if phases is not None:
            if l_out != 2 * srcs[0].length or len(srcs) != 1:
                raise ValueError("explicit phases describe a stride-2 scatter of one source")
            l_cols, cstride, ostride = srcs[0].length, 1, 2
        elif transposed:
            if (taps, stride, pad) != (4, 2, 1) or l_out != 2 * srcs[0].length or len(srcs) != 1:
                raise ValueError("v2 lowers ConvTranspose1d(4, 2, 1) of one source only")
            # (taps in row order, item pad, output offset): even outputs n = 2m read rows m-1, m with taps 3, 1; odd n = 2m+1 rows m, m+1 with 2, 0
            phases = [(w_eff[:, [3, 1], :], 1, 0), (w_eff[:, [2, 0], :], 0, 1)]
            l_cols, cstride, ostride = srcs[0].length, 1, 2
        else:
            if pad > HALO2 or (taps - 1 - pad) > HALO2:
                raise ValueError(f"kernel size {taps} needs more than {HALO2} halo rows")
            phases = [(w_eff, pad, 0)]
            l_cols, cstride, ostride = l_out, stride, 1
            # ---- grouped op (member view of a grouped program): all k trajectories of the group ride the column axis ----
            ex_srcs = [a for ex in (extra or []) for a in ex["srcs"]]
            w_bytes = 4 * (w_eff.numel() + sum(ex["w_eff"].numel() for ex in (extra or [])))
            gop = (self.grouped and self.group_on and self.member[1] > 1 and stride == 1 and bwd is None and save is None and not col_norm
                   and dst.gcap and all(a.gcap and a.length == l_out for a in list(srcs) + ex_srcs) and (res is None or res.gcap)
                   and w_bytes >= group_min_bytes())
            if gop:
                l_cols = self.member[1] * l_out
        if len(phases) != 1 or transposed:
            gop = False
        for w, ppad, _ in phases:
            if ppad > HALO2 or (w.shape[1] - 1 - ppad) > HALO2:
                raise ValueError("phase reaches past the halo rows")
        mode = MODE_4X4 if (l_cols <= 8 and c_out % 64 == 0 and self.allow_4x4) else MODE_16X16
        rows, cols = (16, 16) if mode == MODE_16X16 else (64, 4)
        nt = 2 if (mode == MODE_4X4 and l_cols > 4) else 1
        n_rt = -(-c_out // rows)
        n_cg = -(-l_cols // (nt * cols))
        coutp = pad32(c_out)
        # ---- member view of a split program: which row tiles are this member's, and which lane groups of the epilogue that is ----
        mem, ksp = self.member
        my_rts, xg = list(range(n_rt)), 0
        k_chunks = sum(-(-a.chans // (16 if mode == MODE_16X16 else 4)) for a in srcs) * phases[0][0].shape[1]     # records per row tile
        worth = k_chunks * n_rt * n_cg / self.nw >= SPLIT_MIN_RECORDS
        if gop:
            # the member's 1/k of the row tiles = whole GroupNorm lane groups, or the op stays an ordinary one on the member's own trajectory
            cgw = coutp // GROUPS2
            lo_c, hi_c = mem * n_rt // ksp * rows, (mem + 1) * n_rt // ksp * rows
            # (... and at most two float4 items per epilogue lane: the grouped epilogue of the kernel is instantiated for 1 and 2)
            if n_rt % ksp or n_cg != 1 or coutp != c_out or lo_c % cgw or hi_c % cgw or GROUPS2 % ksp or -(-(cgw // 4 * l_out) // 32) > 2:
                gop, l_cols = False, l_out
                mode = MODE_4X4 if (l_cols <= 8 and c_out % 64 == 0 and self.allow_4x4) else MODE_16X16
                rows, cols = (16, 16) if mode == MODE_16X16 else (64, 4)
                nt = 2 if (mode == MODE_4X4 and l_cols > 4) else 1
                n_rt, n_cg = -(-c_out // rows), -(-l_cols // (nt * cols))
                my_rts = list(range(n_rt))
            else:
                my_rts = list(range(mem * n_rt // ksp, (mem + 1) * n_rt // ksp))
                xg = (lo_c // cgw) | ((hi_c // cgw) << 8) | XG_XCHG | XG_GOP
                self.xchg_floats = max(self.xchg_floats, l_cols * coutp)
                for a in list(srcs) + ex_srcs + [dst] + ([res] if res is not None else []):
                    a.gk = ksp                               # these slots hold the group's k trajectories from now on
        elif ksp > 1 and not self.grouped and self.group_on and worth and len(phases) == 1 and n_rt > 1 and bwd is None and save is None and (n_rt % ksp == 0 or ksp % n_rt == 0):
            cand = list(range(mem * n_rt // ksp, (mem + 1) * n_rt // ksp)) if n_rt >= ksp else [mem * n_rt // ksp]
            cgw = coutp // GROUPS2
            lo_c, hi_c = cand[0] * rows, (cand[-1] + 1) * rows
            if lo_c % cgw == 0 and hi_c % cgw == 0:             # whole GroupNorm lane groups
                my_rts, xg = cand, (lo_c // cgw) | (min(hi_c // cgw, GROUPS2) << 8) | (1 << 16)
                self.xchg_floats = max(self.xchg_floats, l_out * coutp)
        tiles = len(my_rts) * n_cg * len(phases)
        # staged partial tiles: [K slice][output position][channel]; a grouped op stages only the member's channels, for k x l_out columns
        sstride = (len(my_rts) * rows + 4) if gop else coutp + 4
        l_stage = l_cols if gop else l_out
        rt0 = my_rts[0] if gop else 0
        nw, ring = self.nw, ring_depth(self.nw)
        # record stream of a (phase, row tile): [source 0: taps x chunks | source 1: taps x chunks]; a K slice never straddles the
        # sources.  K slices: every source's record range is cut evenly; with two sources (a concat) each source gets at least one
        # slice and the epilogue sums the staged partials.  Phases with fewer taps get the same NUMBER of slices (shorter ones).
        extra = list(extra or [])
        all_srcs = list(srcs)
        kpost, pbias = 0, None
        if len(phases) == 1:
            # ---- single phase: a list of record STREAMS (one per source of the main conv and of every extra conv); every stream is
            # cut into K slices, one work item per (slice, tile).  Slice budget = waves per tile, handed out greedily to the stream with
            # the most records per slice (>= 1 per stream, >= MIN_SLICE records per slice, staging area <= max_stage).
            streams = []                                     # dict(act index, records, ccn, pad, post)
            lo = 0
            for a in srcs:
                r, ccn = _records(phases[0][0][:, :, lo:lo + a.chans], mode)
                streams.append(dict(src=all_srcs.index(a), recs=r, ccn=ccn, pad=phases[0][1], post=False))
                lo += a.chans
            for ex in extra:
                w2, lo = ex["w_eff"], 0
                assert w2.shape[0] == c_out and w2.shape[2] == sum(a.chans for a in ex["srcs"]) and stride == 1
                if ex["pad"] > HALO2 or (w2.shape[1] - 1 - ex["pad"]) > HALO2:
                    raise ValueError("extra conv reaches past the halo rows")
                for a in ex["srcs"]:
                    if a not in all_srcs:
                        all_srcs.append(a)
                    r, ccn = _records(w2[:, :, lo:lo + a.chans], mode)
                    streams.append(dict(src=all_srcs.index(a), recs=r, ccn=ccn, pad=ex["pad"], post=bool(ex.get("post"))))
                    lo += a.chans
                if ex.get("post"):
                    pb = ex.get("bias")
                    pbias = (pbias if pbias is not None else torch.zeros(c_out, device=self.device)) + \
                        (pb.detach().to(self.device) if pb is not None else 0)
                else:
                    assert ex.get("bias") is None, "bias of a summed extra conv: fold it into the main bias"
            budget = max(len(streams), nw // tiles if tiles < nw else 1)
            if gop:      # the post-norm extra streams (1x1 skip) ride as second-round items: the main conv keeps every wave
                budget = (nw // tiles if tiles < nw else 1) + sum(1 for st in streams if st["post"])
                if os.environ.get("CDX_DBG_LONE") == "1":       # (diagnostic: one K slice per tile -- four waves run the steady loop alone on their SIMDs)
                    budget = 1 + sum(1 for st in streams if st["post"])
            per = [1] * len(streams)
            n_of = [st["recs"].shape[1] for st in streams]
            single_round = False
            uneven = {}                                      # stream -> explicit slice ends (see the grouped ops with a skip conv below)
            kpost_streams = [i for i in range(len(streams)) if streams[i]["post"]]

            def grow():
                cand = [i for i in range(len(streams)) if n_of[i] // (per[i] + 1) >= MIN_SLICE]
                return max(cand, key=lambda i: n_of[i] / per[i]) if cand else None
            while sum(per) < budget and (sum(per) + 1) * l_stage * sstride <= self.max_stage:
                i = grow()
                if i is None:
                    break
                per[i] += 1
            if extra and max(n_of[i] / per[i] for i in range(len(srcs))) > self.fuse_max:
                # the main conv is a long, stream-bound K loop: giving waves away to the extra conv costs it more than an op of its own
                return False
            self.macs += sum(c_out * l_out * ex["w_eff"].shape[1] * ex["w_eff"].shape[2] for ex in extra)
            if gop and kpost_streams:
                # A grouped op whose block carries a 1x1 skip conv: its second-round items (the skip streams) start with an empty ring
                # (~1.3 k cycles until their first MFMA) on the waves that also hold a first-round slice.  Cut the main conv so that
                # (i) every slice is a multiple of the ring -- only such slices take the kernel's immediate-offset loop (round 5:
                # op 24G's 13 / 13 / 14 cut ran the general loop at 310 cycles per record) -- and (ii) the slices of the waves that get
                # a second-round item are one ring revolution shorter than the others' (r05 op profile: ops 16G / 24G waited ~3.6 k
                # cycles at their staging barrier for the four waves that drew 48 records instead of 40).
                n_first = nw // tiles if tiles < nw else 1
                main = [i for i in range(len(streams)) if not streams[i]["post"]]
                uneven_on = os.environ.get("CDX_UNET2_UNEVEN_CUT", "1") != "0"
                if uneven_on and len(main) == 1 and n_first - len(kpost_streams) >= 2 and n_of[main[0]] >= 2 * ring * (n_first - len(kpost_streams)):
                    # enough waves per tile for the skip streams to be FIRST-round items next to ring-aligned main slices: one round, no
                    # empty-ring start at all (24G: 24 / 16 main + 16 / 8 skip records per tile instead of 13 / 13 / 14 + a second round)
                    per = [1] * len(streams)
                    per[main[0]] = n_first - len(kpost_streams)
                    single_round = True
                elif uneven_on and len(main) == 1 and n_first == 2 and n_of[main[0]] >= 4 * ring and n_of[main[0]] % (2 * ring) == 0:
                    per[main[0]] = 2
                    uneven[main[0]] = [n_of[main[0]] // 2 - ring, n_of[main[0]]]
                    spare = budget - 2 - sum(per[i] for i in range(len(streams)) if streams[i]["post"])
                    for i in sorted((i for i in range(len(streams)) if streams[i]["post"]), key=lambda i: -n_of[i]):
                        while spare < 0 and per[i] > 1:
                            per[i] -= 1
                            spare += 1
            order = [i for i in range(len(streams)) if not streams[i]["post"]] + [i for i in range(len(streams)) if streams[i]["post"]]
            kpost = sum(per[i] for i in range(len(streams)) if streams[i]["post"])
            ksplit = sum(per) - kpost                        # slices summed BEFORE the norm; the post slices follow them in the stage
            items, item_src, ks = [], [], 0
            for i in order:
                st, n, k = streams[i], n_of[i], per[i]
                woff = self.add(st["recs"].contiguous())
                al = ring if n >= 2 * ring * k else 1
                edge = [min(n, (j * n // k + al // 2) // al * al) for j in range(k)] + [n]
                if i in uneven:
                    edge = [0] + uneven[i]
                for j in range(k):
                    q0, q1 = edge[j], edge[j + 1]
                    for tile in range(n_rt * n_cg):
                        rt, cgi = tile % n_rt, tile // n_rt
                        if rt not in my_rts:
                            continue
                        items.append([woff + (rt * n + q0) * 256, q1 - q0, (q0 // st["ccn"]) | ((q0 % st["ccn"]) << 8),
                                      ks * l_stage * sstride + (rt - rt0) * rows, cgi * nt * cols, st["pad"] | (0 << 8), 0, st["ccn"]])
                        item_src.append(st["src"])
                    ks += 1
            stage_slices = ksplit + kpost
            if single_round and len(items) == nw:
                # waves w and w + 4 share a SIMD: deal the items so that the SIMDs' record totals come out level (largest first onto the
                # least loaded SIMD with a free wave).  An item carries its own tile and partial-sum slot: the order changes no result.
                load, free, place = [0] * 4, [[sm, sm + 4] for sm in range(4)], {}
                for idx in sorted(range(nw), key=lambda t: -items[t][I2_NQ]):
                    sm = min((q for q in range(4) if free[q]), key=lambda q: load[q])
                    place[free[sm].pop(0)] = idx
                    load[sm] += items[idx][I2_NQ]
                items = [items[place[w]] for w in range(nw)]
                item_src = [item_src[place[w]] for w in range(nw)]
        else:
            assert not extra, "extra convs ride on single-phase ops only"
            # record stream of a (phase, row tile): [source 0: taps x chunks | source 1: taps x chunks]; a K slice never straddles the
            # sources.  Phases with fewer taps get the same NUMBER of slices (shorter ones).
            per_src = max(1, (nw // tiles if tiles < nw else 1) // len(srcs))
            ph_segs = []
            for w, _, _ in phases:
                segs, lo = [], 0
                for a in srcs:
                    segs.append(_records(w[:, :, lo:lo + a.chans], mode))
                    lo += a.chans
                ph_segs.append(segs)
            per = [min([per_src] + [max(1, segs[si][0].shape[1] // MIN_SLICE) for segs in ph_segs]) for si in range(len(srcs))]
            while sum(per) > len(srcs) and sum(per) * l_out * sstride > self.max_stage:
                per[per.index(max(per))] -= 1
            ph_info = []
            for (w, ppad, ooff), segs in zip(phases, ph_segs):
                seg_n = [r.shape[1] for r, _ in segs]
                cuts, base = [], 0                          # (first record, one past the last, source index, source base) per slice
                for si, (n, k) in enumerate(zip(seg_n, per)):
                    al = ring if n >= 2 * ring * k else 1
                    edge = [min(n, (j * n // k + al // 2) // al * al) for j in range(k)] + [n]
                    cuts += [(base + edge[j], base + edge[j + 1], si, base) for j in range(k)]
                    base += n
                woff = self.add(torch.cat([r for r, _ in segs], dim=1).contiguous())
                ph_info.append(dict(segs=segs, nqt=sum(seg_n), cuts=cuts, woff=woff, pad=ppad, ooff=ooff))
            ksplit = max(len(pi["cuts"]) for pi in ph_info)
            items, item_src = [], []
            for ks in range(ksplit):
                for tile in range(n_rt * n_cg):
                    rt, cgi = tile % n_rt, tile // n_rt
                    for pi in ph_info:
                        q0, q1, si, sbase = pi["cuts"][ks]
                        ccn = pi["segs"][si][1]
                        items.append([pi["woff"] + (rt * pi["nqt"] + q0) * 256, q1 - q0, ((q0 - sbase) // ccn) | (((q0 - sbase) % ccn) << 8),
                                      ks * l_out * sstride + rt * rows, cgi * nt * cols, pi["pad"] | (pi["ooff"] << 8), 0, ccn])
                        item_src.append(si)
            if any(len(pi["cuts"]) != ksplit for pi in ph_info):
                raise ValueError("phases with different K-slice counts are not supported (stale partial tiles)")
            stage_slices = ksplit
        words = {W2_KIND: KIND2_CONV, W2_COUT: c_out, W2_LOUT: l_out, W2_LCOLS: l_cols, W2_CSTRIDE: cstride, W2_OSTRIDE: ostride,
                 W2_MODE: mode, W2_NT: nt, W2_NITEMS: len(items), W2_DST_STRIDE: dst.stride, W2_SSTRIDE: sstride,
                 W2_KSPLIT: ksplit, W2_COUTP: coutp, W2_KPOST: kpost}
        if bias_row >= 0:
            assert bias is None, "a table-row bias replaces the static one (fold it into the row)"
            words[W2_BOFF] = bias_row
        else:
            words[W2_BOFF] = self.add(_padded(bias if bias is not None else torch.zeros(c_out, device=self.device), coutp))
        if xg:
            words[W2_XG] = xg
        if gop:
            words[W2_GMAP] = (l_out.bit_length() - 1) | ((l_out + 2 * HALO2) << 8)
        if kpost:
            words[W2_PBIAS] = self.add(_padded(pbias, coutp))
        cg = coutp // GROUPS2
        cg4 = cg // 4
        if cg4 & (cg4 - 1) or cg4 > 32:
            raise ValueError(f"C_out {c_out}: the epilogue partitions at most 1024 channels (8 lane groups x 32 float4 lanes)")
        nk = -(-(cg4 * l_out) // 32)
        if nk > MAX_NK2:
            raise ValueError(f"epilogue: {cg4 * l_out} float4 items per group > {32 * MAX_NK2} (horizon too long for v2)")
        words[W2_CG4_SHIFT], words[W2_NK] = cg4.bit_length() - 1, nk
        flags = 0
        reads, writes = list(all_srcs), [dst]
        norm = gn if bwd is None else bwd["gn"]
        if kpost and bwd is not None:
            raise ValueError("post-norm extra convs exist on forward ops only")
        if norm is not None:
            # the epilogue cuts pad32(C_out) channels into 8 lane groups; a real group must be exactly one of them (C_out = 16 with
            # 4 groups of 4 is fine: lane groups 4..7 then work on pad channels that are never stored)
            if norm.num_groups < 1 or c_out % norm.num_groups or c_out // norm.num_groups != cg or abs(norm.eps - GN_EPS) > 1e-12:
                raise ValueError(f"v2 epilogue needs GroupNorm groups of pad32(C)/8 channels (C={c_out}, G={norm.num_groups})")
            flags |= F2_GN if bwd is None else F2_GNBWD
            if col_norm:
                assert bwd is None and cg_real % 4 == 0 and 0 <= cg_real <= cg
                flags |= F2_COLNORM
                words[W2_CGREAL4] = cg_real // 4 if 0 < cg_real < cg else 0
            words[W2_INV_CNT] = _fbits(1.0 / ((cg_real or cg) if col_norm else cg * l_out))
            words[W2_GAMMA], words[W2_BETA] = self.add(_padded(norm.weight, coutp)), self.add(_padded(norm.bias, coutp))
        if bwd is not None:
            assert gn is None and emb_off < 0 and save is None and bwd["save"].chans == c_out and bwd["save"].length == l_out
            words[W2_SAVE_STRIDE], words[W2_STATS] = bwd["save"].stride, bwd["stats"]
            flags |= F2_SAVE_GLOBAL if bwd["save"].in_global else 0
            reads.append(bwd["save"])
            if bwd.get("dst2") is not None:
                assert bwd["dst2"].chans == c_out and bwd["dst2"].length == l_out
                flags |= F2_DUAL
                words[W2_DST2_STRIDE] = bwd["dst2"].stride
                writes.append(bwd["dst2"])
        if save is not None:
            assert gn is not None and save[0].chans == c_out and save[0].length == l_out and save[0].halo == 0
            flags |= F2_SAVE | (F2_SAVE_GLOBAL if save[0].in_global else 0)
            words[W2_SAVE_STRIDE], words[W2_STATS] = save[0].stride, save[1]
            writes.append(save[0])
        if emb_off >= 0:
            flags |= F2_EMB | (F2_FILM if film else 0)
            words[W2_EMB] = emb_off
        if res is not None:
            assert res.chans == c_out and res.length == l_out
            flags |= F2_RES
            words[W2_RES_STRIDE] = res.stride
            reads.append(res)
        if pred:
            flags |= F2_PRED
        if bias_row >= 0:
            flags |= F2_BIAS_EMB
        if act is not None:
            assert bwd is None and 0 <= act < 15
            flags |= (act + 1) << F2_ACT_SHIFT
        if out_div is not None:
            assert bwd is None and save is None
            flags |= F2_OUT_DIV
            words[W2_ODIV] = _fbits(float(out_div))
        words[W2_FLAGS] = flags
        op = [0] * op_words(self.nw)
        for k, v in words.items():
            op[k] = int(v)
        self.ops.append(op)
        self.op_acts.append(dict(srcs=list(all_srcs), res=res, dst=dst, save=(save[0] if save is not None else (bwd["save"] if bwd else None)),
                                 dst2=(bwd.get("dst2") if bwd else None), reads=reads, writes=writes, gop=gop))
        self.op_items.append(items)
        self.op_item_src.append(item_src)
        self.stage = max(self.stage, stage_slices * l_stage * sstride)
        self.macs += sum(c_out * (l_cols if len(phases) > 1 else l_out) * w.shape[1] * c_in for w, _, _ in phases)
        return True

    def head(self, src: Act, dst: Act, w1: torch.Tensor, e_off: int, w2: torch.Tensor, b2: Optional[torch.Tensor] = None):
        """Classifier head with its backward in one op (reference nn_classifier/half_jannerunet.py:49-50, :62):
        z = W1x flat(src) + e (e = W1e emb + b1 comes from the per-step table at `e_off`), y = w2 . Mish(z) (+ b2, irrelevant for the
        gradient); dst <- d y / d src = W1x^T (w2 * Mish'(z)).  `w1` = W1x as [hidden][C][L] (the reference flattens channel-major)."""
        hidden, c, l = w1.shape
        assert (c, l) == (src.chans, src.length) and (dst.chans, dst.length) == (c, l) and w2.numel() == hidden
        if hidden > 256:
            raise ValueError("classifier head wider than 256 hidden units")
        wp = w1.permute(2, 1, 0).contiguous()                # [l][c][hidden]: the thread of hidden unit j reads consecutive j
        words = {W2_KIND: KIND2_HEAD, W2_COUT: hidden, W2_LOUT: l, W2_LCOLS: c, W2_NITEMS: 0, W2_DST_STRIDE: dst.stride,
                 W2_RES_STRIDE: src.stride, W2_BOFF: self.add(wp),
                 # [w2 (hidden) | b2]: the bias only matters for the final log_p forward (cdx_unet2_launch.logp_out)
                 W2_GAMMA: self.add(torch.cat([w2.reshape(-1), (b2 if b2 is not None else w2.new_zeros(1)).reshape(-1)[:1]])), W2_EMB: e_off,
                 W2_COUTP: pad32(c), W2_NK: 1, W2_KSPLIT: 1}
        op = [0] * op_words(self.nw)
        for k, v in words.items():
            op[k] = int(v)
        self.ops.append(op)
        self.head_index = len(self.ops) - 1
        self.op_acts.append(dict(srcs=[], res=src, dst=dst, save=None, dst2=None, reads=[src], writes=[dst]))
        self.op_items.append([])
        self.op_item_src.append([])
        self.stage = max(self.stage, 2 * hidden)
        self.macs += 2 * hidden * c * l

    def load_state(self, dst: Act):
        """Compact guided programs: the state x_t is read back from GLOBAL memory (the launch's x_out, where compact programs keep
        it) into a fresh slot for the classifier's first ops -- the denoiser's own copy need not stay in LDS across its peak."""
        words = {W2_KIND: KIND2_LOADX, W2_COUT: dst.chans, W2_LOUT: dst.length, W2_NITEMS: 0, W2_DST_STRIDE: dst.stride,
                 W2_COUTP: pad32(dst.chans), W2_NK: 1, W2_KSPLIT: 1}
        op = [0] * op_words(self.nw)
        for k, v in words.items():
            op[k] = int(v)
        self.ops.append(op)
        self.op_acts.append(dict(srcs=[], res=None, dst=dst, save=None, dst2=None, reads=[], writes=[dst]))
        self.op_items.append([])
        self.op_item_src.append([])

    def load_context(self, dst: Act):
        """Batch-tiled MLP programs: the per-sample condition features of the workgroup's samples (launch `ctx`, (batch, tile, C)) ->
        slot `dst` at the start of every forward; zeros for the unconditional forward of a classifier-free-guidance pair and when the
        request has no condition (the reference substitutes zeros, e.g. pearcemlp.py:59-60)."""
        words = {W2_KIND: KIND2_LOADC, W2_COUT: dst.chans, W2_LOUT: dst.length, W2_NITEMS: 0, W2_DST_STRIDE: dst.stride,
                 W2_COUTP: pad32(dst.chans), W2_NK: 1, W2_KSPLIT: 1}
        op = [0] * op_words(self.nw)
        for k, v in words.items():
            op[k] = int(v)
        self.ops.append(op)
        self.op_acts.append(dict(srcs=[], res=None, dst=dst, save=None, dst2=None, reads=[], writes=[dst]))
        self.op_items.append([])
        self.op_item_src.append([])

    def plan_arena(self, base: int) -> int:
        """Interval allocation of the non-persistent slots over op liveness; patches slot offsets into the ops.  `alias_residual`:
        the output of an op whose residual slot has the same shape and dies with this op is written IN PLACE over that residual
        (each epilogue thread reads its residual element before it stores the same element) -- one slot less at the deepest level."""
        first, last = {}, {}
        for i, oa in enumerate(self.op_acts):
            for a in oa["reads"] + oa["writes"]:
                first.setdefault(a.uid, i)
                last[a.uid] = i
        for a in self.keep_to_end:
            last[a.uid] = len(self.op_acts) - 1
        root = {}                                            # aliased slot uid -> uid of the slot whose storage it shares
        if self.alias_residual:
            for i, oa in enumerate(self.op_acts):
                r, d = oa["res"], oa["dst"]
                if (r is not None and r is not d and not r.persistent and not d.persistent and r.uid not in root
                        and (r.length, r.chans, r.halo) == (d.length, d.chans, d.halo) and last[r.uid] == i and first[d.uid] == i
                        and int(self.ops[i][W2_KIND]) == KIND2_CONV and not (int(self.ops[i][W2_FLAGS]) & (F2_GNBWD | F2_DUAL))):
                    root[d.uid] = r.uid
                    last[r.uid] = max(last[r.uid], last[d.uid])
        slots = [a for a in self.acts if not a.persistent and a.uid in first and a.uid not in root]
        n_ops = len(self.op_acts)
        wrap = {a.uid for a in self.wrap_live}               # slots written AFTER the last op (solver step) and read by op 0

        def overlap(a, b):
            if not (last[b.uid] < first[a.uid] or last[a.uid] < first[b.uid]):
                return True
            for w, o in ((a, b), (b, a)):                    # a wrap slot is also live at the last op (the solver step reads `pred` then)
                if w.uid in wrap and last[o.uid] >= n_ops - 1:
                    return True
            return False

        def place(order):
            offs, placed = {}, []
            for a in order:
                busy = sorted((offs[b.uid], offs[b.uid] + b.floats) for b in placed if overlap(a, b))
                pos = base
                for lo, hi in busy:
                    if lo - pos >= a.floats:
                        break
                    pos = max(pos, hi)
                offs[a.uid] = pos
                placed.append(a)
            return offs, max([base] + [offs[a.uid] + a.floats for a in placed])
        # two orders, keep the tighter plan: by first use (what a stack would do) and largest-first
        best = min((place(sorted(slots, key=lambda a: first[a.uid])), place(sorted(slots, key=lambda a: (-a.floats, first[a.uid])))),
                   key=lambda r: r[1])
        offs, top = best
        by_uid = {a.uid: a for a in self.acts}
        for a in slots:
            a.off = offs[a.uid]
        for d, r in root.items():
            by_uid[d].off = by_uid[r].off
        mem = self.member[0]
        for op, oa, items, item_src in zip(self.ops, self.op_acts, self.op_items, self.op_item_src):
            # grouped programs: an ORDINARY op works on the member's own trajectory = sub-slot `mem` of every group slot it touches
            # (a grouped op addresses sub-slots itself: column -> trajectory)
            own = lambda a: a.off + (mem * a.sub_floats if (a.gk > 1 and not oa.get("gop")) else 0)      # noqa: E731
            op[W2_DST] = own(oa["dst"])                                # slot base = first halo row
            if oa["res"] is not None:
                op[W2_RES] = own(oa["res"])
            if oa["save"] is not None:
                op[W2_SAVE] = oa["save"].off
            if oa["dst2"] is not None:
                op[W2_DST2] = oa["dst2"].off
            for rec, si in zip(items, item_src):
                src = oa["srcs"][si]
                if own(src) >= (1 << 16) or src.stride >= (1 << 15):     # item words pack offset | stride << 16
                    raise ValueError(f"LDS plan beyond the 16-bit slot offsets of the item format (slot at float {src.off})")
                rec[I2_SRCSTR] = own(src) | (src.stride << 16)
            if oa["dst"].gk > 1 and not oa.get("gop"):
                # an ordinary op wrote its trajectory into a group slot that grouped ops read: the members gather whole trajectories
                d = oa["dst"]
                if int(op[W2_KIND]) != KIND2_CONV or oa["res"] is d or (int(op[W2_FLAGS]) & (F2_GNBWD | F2_SAVE | F2_DUAL)):
                    raise ValueError("grouped program: only plain forward convs may write a group slot from one trajectory")
                op[W2_XG] = 0 | (GROUPS2 << 8) | XG_XCHG | XG_TRAJ
                op[W2_GMAP] = (d.length.bit_length() - 1) | ((d.length + 2 * HALO2) << 8)
                self.xchg_floats = max(self.xchg_floats, d.gk * d.length * int(op[W2_COUTP]))
        return top


def op_item(ops_buffer: np.ndarray, op: np.ndarray, j: int) -> np.ndarray:
    """Work item `j` of `op` (a row of Program2.ops): inline in the descriptor for j < nw, else in the tail table."""
    nw = (len(op) - HDR_WORDS) // ITEM2_WORDS
    if j < nw:
        return op[W2_ITEM0 + j * ITEM2_WORDS: W2_ITEM0 + (j + 1) * ITEM2_WORDS]
    lo = int(op[W2_ITEMS]) + (j - nw) * ITEM2_WORDS
    return ops_buffer[lo: lo + ITEM2_WORDS]


def _padded(v: torch.Tensor, n: int) -> torch.Tensor:
    v = v.detach().to(torch.float32).reshape(-1)
    return torch.cat([v, torch.zeros(n - v.numel(), device=v.device)]) if v.numel() < n else v


def _emb_table_spec(b: "_Builder2", net, blocks, dev, raw_rows: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None) -> dict:
    """Blob offsets of a network's embedding MLP for cdx_unet2_embtab: map_emb (Linear -> Mish -> Linear), then per block
    emb_mlp = Mish -> Linear stacked into one matrix over Mish(map_emb(temb)); `raw_rows` = (W, bias, table offset): rows applied to
    the RAW map_emb output (the classifier head's share of its first Linear)."""
    emb = {"emb_dim": net.emb_dim, "hidden": net.map_emb[0].out_features, "md": net.map_emb[2].out_features}
    emb["w0"], emb["b0"] = b.add(net.map_emb[0].weight.t().contiguous()), b.add(net.map_emb[0].bias)
    emb["w2"], emb["b2"] = b.add(net.map_emb[2].weight.t().contiguous()), b.add(net.map_emb[2].bias)
    lo = min(off for _, off in blocks)
    hi = max(off + pad32(rb.emb_mlp[1].out_features) for rb, off in blocks)
    w_all = torch.zeros(hi - lo, emb["md"], device=dev)
    b_all = torch.zeros(hi - lo, device=dev)
    for rb, off in blocks:
        lin = rb.emb_mlp[1]
        w_all[off - lo:off - lo + lin.out_features] = lin.weight.detach().to(dev)
        b_all[off - lo:off - lo + lin.out_features] = lin.bias.detach().to(dev)
    emb["w3"], emb["b3"], emb["col0"], emb["n_emb"] = b.add(w_all.t().contiguous()), b.add(b_all), lo, hi - lo
    emb["w4"], emb["b4"], emb["col4"], emb["n_raw"] = 0, 0, 0, 0
    if raw_rows is not None:
        w, bias, col = raw_rows
        emb["w4"], emb["b4"], emb["col4"], emb["n_raw"] = b.add(w.t().contiguous()), b.add(bias), col, w.shape[0]
    b.macs += net.emb_dim * emb["hidden"] + emb["hidden"] * emb["md"] + emb["md"] * sum(rb.emb_mlp[1].out_features for rb, _ in blocks)
    return emb


def _lower_janner(b: "_Builder2", net, horizon: int, x: Act):
    """Ops of one JannerUNet1d forward (reference nn_diffusion/jannerunet.py:154-201) reading slot `x`; returns (pred slot, blocks)."""
    d, k, md = net.in_dim, net.kernel_size, net.model_dim
    blocks = []

    def resblock(srcs: List[Act], rb) -> Act:
        """ResidualBlock (jannerunet.py:51-69) over the channel concat of `srcs`."""
        c_out, length = rb.conv1[0].out_channels, srcs[0].length
        e_off = b.emb_slot(c_out)
        blocks.append((rb, e_off))
        t1 = b.act(length, c_out)
        b.conv(srcs, t1, _conv1d_eff(rb.conv1[0]), rb.conv1[0].bias, pad=k // 2, gn=rb.conv1[1], emb_off=e_off)
        out = b.act(length, c_out)
        if isinstance(rb.residual_conv, nn.Identity):
            assert len(srcs) == 1
            b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1], res=srcs[0])
        else:
            # the 1x1 skip conv rides in the second conv's op when that pays: its work items take some of the waves, its partial tiles
            # are added after the norm / activation (one op, two barriers and one epilogue less per block)
            fused = FUSE_SKIP and b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1], extra=[
                dict(srcs=srcs, w_eff=_conv1d_eff(rb.residual_conv), pad=0, bias=rb.residual_conv.bias, post=True)])
            if not fused:
                b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1])
                b.conv(srcs, out, _conv1d_eff(rb.residual_conv), rb.residual_conv.bias, res=out)     # out += W_r x + b_r
        return out

    cur, skips = x, []
    for res1, res2, _, down in net.downs:
        cur = resblock([resblock([cur], res1)], res2)
        skips.append(cur)
        if not isinstance(down, nn.Identity):
            if cur.length % 2:
                raise ValueError("horizon too short for the number of resolutions")
            nxt = b.act((cur.length - 1) // 2 + 1, cur.chans)
            b.conv([cur], nxt, _conv1d_eff(down.conv), down.conv.bias, stride=2, pad=1)
            cur = nxt
    cur = resblock([resblock([cur], net.mid_block1)], net.mid_block2)
    for res1, res2, _, up in net.ups:
        cur = resblock([resblock([cur, skips.pop()], res1)], res2)
        if not isinstance(up, nn.Identity):
            nxt = b.act(cur.length * 2, cur.chans)
            b.conv([cur], nxt, _convT1d_eff(up.conv), up.conv.bias, stride=2, pad=1, transposed=True)
            cur = nxt
    if cur.length != horizon:
        raise ValueError("up path does not return to the input horizon")
    fc = net.final_conv
    if cur.chans != fc[0].in_channels:
        # dim_mult[0] != 1: the reference's own forward fails on this net (jannerunet.py:139-144 builds final_conv on model_dim)
        raise ValueError(f"final conv expects {fc[0].in_channels} channels, the up path ends with {cur.chans}")
    t = b.act(horizon, md)
    b.conv([cur], t, _conv1d_eff(fc[0]), fc[0].bias, pad=2, gn=fc[1])
    return t, fc, blocks


def _lower_chiunet(b: "_Builder2", net, horizon: int, x: Act):
    """Ops of one ChiUNet1d forward with a global condition (reference nn_diffusion/chiunet.py:152-192) reading slot `x`; returns
    (last hidden slot, final_conv, [(block, table offset)]).  A block's FiLM vector Linear(Mish(emb)) (chiunet.py:36-45) is a row of
    the launch's per-(step, trajectory) table, like JannerUNet1d's time vectors; with cond_predict_scale the row holds [scale | bias]."""
    k = net.final_conv[0].kernel_size[0]
    blocks = []

    def resblock(srcs: List[Act], rb) -> Act:
        c_out, length = rb.out_dim, srcs[0].length
        film = bool(rb.cond_predict_scale)
        e_off = b.emb_slot(c_out, film)
        blocks.append((rb, e_off))
        t1 = b.act(length, c_out)
        b.conv(srcs, t1, _conv1d_eff(rb.conv1[0]), rb.conv1[0].bias, pad=k // 2, gn=rb.conv1[1], emb_off=e_off, film=film)
        out = b.act(length, c_out)
        if isinstance(rb.residual_conv, nn.Identity):
            assert len(srcs) == 1
            b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1], res=srcs[0])
        else:
            fused = FUSE_SKIP and b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1], extra=[
                dict(srcs=srcs, w_eff=_conv1d_eff(rb.residual_conv), pad=0, bias=rb.residual_conv.bias, post=True)])
            if not fused:
                b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1])
                b.conv(srcs, out, _conv1d_eff(rb.residual_conv), rb.residual_conv.bias, res=out)
        return out

    cur, skips = x, []
    for res1, res2, down in net.downs:
        cur = resblock([resblock([cur], res1)], res2)
        skips.append(cur)
        if not isinstance(down, nn.Identity):
            if cur.length % 2:
                raise ValueError("horizon too short for the number of resolutions")
            nxt = b.act((cur.length - 1) // 2 + 1, cur.chans)
            b.conv([cur], nxt, _conv1d_eff(down.conv), down.conv.bias, stride=2, pad=1)
            cur = nxt
    for mid in net.mids:
        cur = resblock([cur], mid)
    for res1, res2, up in net.ups:
        cur = resblock([resblock([cur, skips.pop()], res1)], res2)
        if not isinstance(up, nn.Identity):
            nxt = b.act(cur.length * 2, cur.chans)
            b.conv([cur], nxt, _convT1d_eff(up.conv), up.conv.bias, stride=2, pad=1, transposed=True)
            cur = nxt
    if cur.length != horizon:
        raise ValueError("up path does not return to the input horizon")
    fc = net.final_conv
    t = b.act(horizon, net.model_dim)
    b.conv([cur], t, _conv1d_eff(fc[0]), fc[0].bias, pad=k // 2, gn=fc[1])
    return t, fc, blocks


def chiunet_film_spec(net, blocks, n_emb: int, dev) -> dict:
    """What the host needs to fill the FiLM table of a ChiUNet1d program (runtime2.chi_film_table): the blocks' cond_encoder Linears
    stacked in table-column order and split by input half -- row(step, trajectory) = W_t Mish(map_emb(temb_step)) + W_c
    Mish(global_cond_encoder(cond_trajectory)) + bias (reference chiunet.py:160-163, :36: emb = [time | condition], every block
    consumes Mish(emb)).  Pad columns (pad32) stay zero: scale 0, bias 0 on channels that are never stored."""
    e = net.emb_dim
    w_t, w_c = torch.zeros(n_emb, e, device=dev), torch.zeros(n_emb, e, device=dev)
    bias = torch.zeros(n_emb, device=dev)
    for rb, off in blocks:
        lin, c, cp = rb.cond_encoder[1], rb.out_dim, pad32(rb.out_dim)
        if lin.in_features != 2 * e:
            raise ValueError("ChiUNet1d block whose FiLM input is not [time emb | encoded condition]")
        w, bv = lin.weight.detach().to(dev, torch.float32), lin.bias.detach().to(dev, torch.float32)
        for part in range(2 if rb.cond_predict_scale else 1):       # [scale | bias] halves sit pad32(C) apart in the table
            rows = slice(off + part * cp, off + part * cp + c)
            w_t[rows], w_c[rows], bias[rows] = w[part * c:(part + 1) * c, :e], w[part * c:(part + 1) * c, e:], bv[part * c:(part + 1) * c]
    return {"w_t": w_t.contiguous(), "w_c": w_c.contiguous(), "bias": bias.contiguous()}


def supports_chiunet2(net) -> Optional[str]:
    if not net.obs_as_global_cond:
        return "local (per-timestep) observation conditioning: implicit-GEMM executor only"
    k = net.final_conv[0].kernel_size[0]
    if k % 2 == 0 or k // 2 > HALO2:
        return f"kernel_size={k} unsupported (odd, <= {2 * HALO2 + 1})"
    return None


def compile_chiunet2(net, horizon: int, max_lds_bytes: int = 160 * 1024, allow_4x4: bool = True, nw: int = NW2, compact: bool = False,
                     max_stage: Optional[int] = None) -> Program2:
    """Lower a ChiUNet1d with a global condition (reference nn_diffusion/chiunet.py:48-192) for `horizon` positions; same program
    format and kernel as JannerUNet1d.  The FiLM table is filled by the host from ``meta["chi_film"]`` (no embedding MLP spec)."""
    why = supports_chiunet2(net)
    if why is not None:
        raise ValueError(why)
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    b.allow_4x4 = allow_4x4
    b.alias_residual = compact
    if max_stage is not None:
        b.max_stage = max_stage
    d = net.final_conv[3].out_channels
    x = b.act(horizon, d, persistent=True)
    t, fc, blocks = _lower_chiunet(b, net, horizon, x)
    pred = b.act(horizon, d)
    b.conv([t], pred, _conv1d_eff(fc[3]), fc[3].bias, pred=True)
    e = net.emb_dim
    b.macs += e * 4 * e + 4 * e * e + net.global_cond_encoder.in_features * e + 2 * e * sum(rb.cond_encoder[1].out_features for rb, _ in blocks)
    prog = _finalize2(b, [{}], x, pred, horizon, d, e, max_lds_bytes, [], compact=compact)
    prog.embtabs = []
    prog.meta["chi_film"] = chiunet_film_spec(net, blocks, b.n_emb, dev)
    prog.meta["cond_dim"] = net.global_cond_encoder.in_features
    return prog


# ================================================================================================== #
# Batch-tiled MLP denoisers on the v2 kernel: a "trajectory" is a tile of samples, the sample index   #
# rides the position axis, a Linear is a 1-tap conv.  Everything that depends on the timestep only    #
# (time embeddings through their MLPs, the raw timestep column) is folded by the HOST into per-step   #
# bias rows of the launch's table; the per-sample condition sits in a context slot reloaded at the    #
# start of every forward (zeros for the unconditional forward of a classifier-free-guidance pair).    #
# ================================================================================================== #
class _RowSpec:
    """Table-row bookkeeping of an MLP program: row[off : off + n] = W feat(src) + bias with src in {"temb": map_noise(t),
    "tfeat": the net's own time MLP over it, "t": the raw timestep}."""

    def __init__(self, b: "_Builder2"):
        self.b, self.terms = b, []

    def row(self, c_out: int, bias: Optional[torch.Tensor], **parts) -> int:
        off = self.b.emb_slot(c_out)
        self.terms.append((off, c_out, bias, parts))
        return off

    def finish(self, n_emb: int, dev) -> dict:
        out = {"bias": torch.zeros(n_emb, device=dev)}
        for off, n, bias, parts in self.terms:
            if bias is not None:
                out["bias"][off:off + n] = bias.detach().to(dev, torch.float32)
            for src, w in parts.items():
                w = w.detach().to(dev, torch.float32).reshape(n, -1)
                m = out.get(src)
                if m is None:
                    m = out[src] = torch.zeros(n_emb, w.shape[1], device=dev)
                m[off:off + n] = w
        return out


def _lin_eff2(w: torch.Tensor) -> torch.Tensor:
    """(n_out, n_in) Linear weight (slice) -> [co][tap = 1][ci]."""
    return w.detach().unsqueeze(1)


def _finish_mlp(b: "_Builder2", rows: _RowSpec, kind: str, x: Act, pred: Act, ctx: Optional[Act], tile: int, d: int, emb_dim: int,
                max_lds_bytes: int, dev) -> Program2:
    prog = _finalize2(b, [{}], x, pred, tile, d, emb_dim, max_lds_bytes, [ctx] if ctx is not None else [])
    prog.embtabs = []
    prog.meta["mlp"] = dict(kind=kind, tile=tile, cond_dim=0 if ctx is None else ctx.chans, rows=rows.finish(b.n_emb, dev))
    return prog


def compile_pearce_mlp2(net, tile: int, max_lds_bytes: int = 160 * 1024, nw: int = NW2_MAX) -> Program2:
    """PearceMlp (reference nn_diffusion/pearcemlp.py:36-79).  fcs[0] reads [act_emb(x) | map_noise(t) | condition]: the time part
    becomes its bias row; fcs[1..3] read [skip | x | raw t]: the t column (times the timestep) joins their bias rows.  FCBlock =
    Linear -> per-sample GroupNorm -> GELU(erf); the skips are stored divided by 1.414 exactly where the reference divides (Q11).

    Hidden widths whose GroupNorm groups are not a power-of-two number of float4 lanes (hidden_dim 192: 8 groups of 24 channels) get a
    PADDED hidden layout: group g sits at channels [G g, G g + 24) of a 8 G-wide slot (G = 32), the pad channels have zero weights, bias,
    gamma and beta -- they hold exact zeros through norm, GELU, skip and division -- and stay out of the variance (W2_CGREAL4)."""
    from .consts import ACT_GELU_ERF, ACT_LEAKY, ACT_NONE
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    rows = _RowSpec(b)
    d, e, hd, n_cond = net.act_dim, net.emb_dim, net.hidden_dim, net.To * net.emb_dim
    gn0 = net.fcs[0].model[1]
    n_grp = gn0.num_groups
    if hd % n_grp or (hd // n_grp) % 4:
        raise ValueError(f"PearceMlp hidden_dim {hd}: GroupNorm groups of {hd / n_grp:g} channels (a multiple of 4 needed)")
    cg_real = hd // n_grp
    cgl = 4
    while cgl < cg_real:
        cgl *= 2
    hp = cgl * GROUPS2 if (cgl != cg_real or n_grp != GROUPS2) else hd          # slot width of the hidden activations
    if n_grp > GROUPS2 or pad32(hp) != hp:
        raise ValueError(f"PearceMlp hidden_dim {hd} with {n_grp} GroupNorm groups has no epilogue partition")
    cmap = torch.tensor([cgl * g + j for g in range(n_grp) for j in range(cg_real)], device=dev)      # real channel -> slot channel

    def out_rows(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:       # (hd, ...) -> (hp, ...), pad rows zero
        if t is None or hp == hd:
            return t
        o = torch.zeros((hp,) + tuple(t.shape[1:]), device=dev, dtype=torch.float32)
        o[cmap] = t.detach().to(dev, torch.float32)
        return o

    def in_cols(w: torch.Tensor) -> torch.Tensor:                            # (n, hd) -> (n, hp), pad columns zero
        if hp == hd:
            return w
        o = torch.zeros(w.shape[0], hp, device=dev, dtype=torch.float32)
        o[:, cmap] = w.detach().to(dev, torch.float32)
        return o

    class _Norm:                                                             # the GroupNorm of a block in the slot's channel layout
        def __init__(self, gn):
            self.num_groups, self.eps = GROUPS2 if hp != hd else gn.num_groups, gn.eps
            self.weight, self.bias = out_rows(gn.weight.detach()), out_rows(gn.bias.detach())
    creal = cg_real if hp != hd else 0
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, n_cond, persistent=True)
    b.load_context(ctx)
    a1, xe = b.act(tile, e), b.act(tile, e)
    b.conv([x], a1, _lin_eff2(net.act_emb[0].weight), net.act_emb[0].bias, act=ACT_LEAKY)
    b.conv([a1], xe, _lin_eff2(net.act_emb[2].weight), net.act_emb[2].bias, act=ACT_NONE)
    f = net.fcs
    w0 = f[0].model[0].weight.detach()
    h1 = b.act(tile, hp)
    b.conv([xe, ctx], h1, _lin_eff2(out_rows(torch.cat([w0[:, :e], w0[:, 2 * e:]], 1))), None, gn=_Norm(f[0].model[1]), col_norm=True,
           act=ACT_GELU_ERF, bias_row=rows.row(hp, out_rows(f[0].model[0].bias), temb=out_rows(w0[:, e:2 * e])), out_div=net.SKIP_SCALE,
           cg_real=creal)
    cur = h1
    for i in (1, 2):
        w = f[i].model[0].weight.detach()
        nxt = b.act(tile, hp)
        b.conv([cur, x], nxt, _lin_eff2(out_rows(torch.cat([in_cols(w[:, :hd]), w[:, hd:hd + d]], 1))), None, gn=_Norm(f[i].model[1]),
               col_norm=True, act=ACT_GELU_ERF, res=cur, cg_real=creal,
               bias_row=rows.row(hp, out_rows(f[i].model[0].bias), t=out_rows(w[:, hd + d:])), out_div=net.SKIP_SCALE if i == 1 else None)
        cur = nxt
    pred = b.act(tile, d)
    w3 = f[3].weight.detach()
    b.conv([cur, x], pred, _lin_eff2(torch.cat([in_cols(w3[:, :hd]), w3[:, hd:hd + d]], 1)), None, pred=True, act=ACT_NONE,
           bias_row=rows.row(d, f[3].bias, t=w3[:, hd + d:]))
    return _finish_mlp(b, rows, "pearce", x, pred, ctx, tile, d, e, max_lds_bytes, dev)


def compile_dql_mlp2(net, tile: int, max_lds_bytes: int = 160 * 1024, nw: int = NW2_MAX) -> Program2:
    """DQLMlp (reference nn_diffusion/dqlmlp.py:9-52) and DVInvMlp (dvinvmlp.py:9-47, same trunk): features [x | time_mlp(map_noise(t))
    | obs] -> 3 x (Linear, Mish) -> Linear; the time features are batch-invariant: they enter as the first layer's bias row."""
    from .consts import ACT_MISH, ACT_NONE
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    rows = _RowSpec(b)
    d, e, obs = net.final_layer.out_features, net.time_mlp[0].in_features, net.obs_dim
    m = net.mid_layer
    hid = m[0].out_features
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, obs, persistent=True)
    b.load_context(ctx)
    w0 = m[0].weight.detach()
    m1, m2, m3 = b.act(tile, hid), b.act(tile, hid), b.act(tile, hid)
    b.conv([x, ctx], m1, _lin_eff2(torch.cat([w0[:, :d], w0[:, d + e:]], 1)), None, act=ACT_MISH,
           bias_row=rows.row(hid, m[0].bias, tfeat=w0[:, d:d + e]))
    b.conv([m1], m2, _lin_eff2(m[2].weight), m[2].bias, act=ACT_MISH)
    b.conv([m2], m3, _lin_eff2(m[4].weight), m[4].bias, act=ACT_MISH)
    pred = b.act(tile, d)
    b.conv([m3], pred, _lin_eff2(net.final_layer.weight), net.final_layer.bias, pred=True, act=ACT_NONE)
    return _finish_mlp(b, rows, "dql", x, pred, ctx, tile, d, e, max_lds_bytes, dev)


def compile_mlp_nn2(net, tile: int, max_lds_bytes: int = 160 * 1024, nw: int = NW2_MAX) -> Program2:
    """MlpNNDiffusion (reference nn_diffusion/mlps.py:10-40): Mlp(cat[x, map_noise(t) + condition]); the first Linear's embedding
    columns act on the time embedding (bias row) and on the condition (context slot) alike."""
    from .consts import _act_id
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    rows = _RowSpec(b)
    layers = list(net.mlp.mlp)
    lins = [m[0] if isinstance(m, nn.Sequential) else m for m in layers if isinstance(m, (nn.Sequential, nn.Linear))]
    acts = [_act_id(m[1]) for m in layers if isinstance(m, nn.Sequential)] + [_act_id(layers[-1])]
    if any(a is None for a in acts) or len(lins) != len(acts):
        raise ValueError("MlpNNDiffusion: activation without a native epilogue")
    d = lins[-1].out_features
    e = lins[0].in_features - d
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, e, persistent=True)
    b.load_context(ctx)
    cur, pred = None, None
    for i, (lin, act) in enumerate(zip(lins, acts)):
        last = i == len(lins) - 1
        dst = b.act(tile, lin.out_features)
        if i == 0:
            w0 = lin.weight.detach()
            b.conv([x, ctx], dst, _lin_eff2(w0), None, act=act, pred=last, bias_row=rows.row(lin.out_features, lin.bias, temb=w0[:, d:]))
        else:
            b.conv([cur], dst, _lin_eff2(lin.weight), lin.bias, act=act, pred=last)
        cur = dst
    pred = cur
    return _finish_mlp(b, rows, "mlpnn", x, pred, ctx, tile, d, e, max_lds_bytes, dev)


def compile_sfbc_unet2(net, tile: int, max_lds_bytes: int = 160 * 1024, nw: int = NW2_MAX) -> Program2:
    """SfBCUNet (reference nn_diffusion/sfbc_unet.py:9-82): block(x, c) = SiLU(L2(SiLU(L1 x) + Lc c)) + skip(x) with c =
    t_layer(map_noise(t)) + condition.  Lc t_layer(...) + bc is a per-step vector added after the first activation (table row);
    Lc condition and the skip Linear are 1-tap convs whose partial tiles are added after the activation of the op they ride in."""
    from .consts import ACT_NONE, ACT_SILU
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    b.fuse_max = 1 << 30                  # (Linears: the extra streams are what the block IS, not an optimisation)
    rows = _RowSpec(b)
    d, e = net.out_layer.out_features, net.t_layer[0].in_features
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, e, persistent=True)
    b.load_context(ctx)

    def block(srcs: List[Act], blk) -> Act:
        c_out = blk.linear1[0].out_features
        h, o = b.act(tile, c_out), b.act(tile, c_out)
        wc = blk.linearc.weight.detach()
        ok = b.conv(srcs, h, _lin_eff2(blk.linear1[0].weight), blk.linear1[0].bias, act=ACT_SILU,
                    emb_off=rows.row(c_out, blk.linearc.bias, tfeat=wc),
                    extra=[dict(srcs=[ctx], w_eff=_lin_eff2(wc), pad=0, bias=None, post=True)])
        if not ok:
            raise ValueError("SfBCUNet block too wide for one op")
        if isinstance(blk.skip, nn.Identity):
            if len(srcs) == 1:
                b.conv([h], o, _lin_eff2(blk.linear2[0].weight), blk.linear2[0].bias, act=ACT_SILU, res=srcs[0])
                return o
            skip_w, skip_b = torch.eye(c_out, device=dev), None          # identity skip over a concat whose widths happen to match
        else:
            skip_w, skip_b = blk.skip.weight.detach(), blk.skip.bias
        ok = b.conv([h], o, _lin_eff2(blk.linear2[0].weight), blk.linear2[0].bias, act=ACT_SILU,
                    extra=[dict(srcs=srcs, w_eff=_lin_eff2(skip_w), pad=0, bias=skip_b, post=True)])
        if not ok:
            raise ValueError("SfBCUNet block too wide for one op")
        return o

    cur, kept = x, []
    for blk in net.down_blocks:
        cur = block([cur], blk)
        kept.append(cur)
    cur = block([cur], net.mid_block)
    for blk in net.up_blocks:
        cur = block([cur, kept.pop()], blk)
    pred = b.act(tile, d)
    b.conv([cur], pred, _lin_eff2(net.out_layer.weight), net.out_layer.bias, pred=True, act=ACT_NONE)
    return _finish_mlp(b, rows, "sfbc", x, pred, ctx, tile, d, e, max_lds_bytes, dev)


MLP2_COMPILERS = {"pearce": compile_pearce_mlp2, "dql": compile_dql_mlp2, "mlpnn": compile_mlp_nn2, "sfbc": compile_sfbc_unet2}


def _dgrad_eff(conv: nn.Conv1d) -> torch.Tensor:
    """Weights of the backward-data conv of a stride-1 'same' Conv1d as a forward conv: [C_in][tap'][C_out], tap' = k-1-tap."""
    return conv.weight.detach().permute(1, 2, 0).flip(1)


def _lower_half_janner_grad(b: "_Builder2", clf, horizon: int, x: Act, grad: Optional[Act], forward_only: bool = False):
    """Forward AND backward-data pass of a HalfJannerUNet1d classifier (reference nn_classifier/half_jannerunet.py:102-125) reading
    slot `x`: d out / d x -> slot `grad` (what BaseClassifier.gradients returns, classifier/base.py:74-79, for the summed log p).
    Every GroupNorm layer saves its normalised values + rstd on the way up; on the way down each backward-data conv's epilogue applies
    the backward of the (GroupNorm -> Mish) below it.  Returns the (block, emb offset) list and the head's table offset.
    `forward_only`: the forward ops and the head only, nothing saved (the classifier's own program: log p of a batch)."""
    if clf.norm_type != "groupnorm" or clf.out_dim != 1:
        raise ValueError("the fused classifier gradient needs norm_type='groupnorm' and out_dim == 1")
    if horizon != clf.horizon:
        raise ValueError(f"HalfJannerUNet1d was built for horizon {clf.horizon}, got {horizon}")
    k = clf.kernel_size
    blocks, tape = [], []                 # tape: forward records consumed in reverse

    def resblock(src: Act, rb, ksz: int) -> Act:
        c_out, length = rb.conv1[0].out_channels, src.length
        e_off = b.emb_slot(c_out)
        blocks.append((rb, e_off))
        s1, s2 = (None, None) if forward_only else ((b.save_slot(length, c_out), b.stats_slot()), (b.save_slot(length, c_out), b.stats_slot()))
        t1 = b.act(length, c_out)
        b.conv([src], t1, _conv1d_eff(rb.conv1[0]), rb.conv1[0].bias, pad=ksz // 2, gn=rb.conv1[1], emb_off=e_off, save=s1)
        out = b.act(length, c_out)
        ident = isinstance(rb.residual_conv, nn.Identity)
        fused = (not ident) and FUSE_SKIP and b.conv(
            [t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=ksz // 2, gn=rb.conv2[1], save=s2,
            extra=[dict(srcs=[src], w_eff=_conv1d_eff(rb.residual_conv), pad=0, bias=rb.residual_conv.bias, post=True)])
        if not fused:
            b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=ksz // 2, gn=rb.conv2[1], res=src if ident else None, save=s2)
            if not ident:
                b.conv([src], out, _conv1d_eff(rb.residual_conv), rb.residual_conv.bias, res=out)
        tape.append(("block", rb, ksz, src, s1, s2, ident))
        return out

    def down(src: Act, dn) -> Act:
        if src.length % 2:
            raise ValueError("horizon too short for the classifier's resolutions")
        nxt = b.act(src.length // 2, src.chans)
        b.conv([src], nxt, _conv1d_eff(dn.conv), dn.conv.bias, stride=2, pad=1)
        tape.append(("down", dn, src))
        return nxt

    cur = x
    for res1, res2, dn in clf.downs:
        cur = resblock(resblock(cur, res1, k), res2, k)
        if not isinstance(dn, nn.Identity):
            cur = down(cur, dn)
    for blk, dn in (clf.mid_block1, clf.mid_block2):
        cur = down(resblock(cur, blk, 5), dn)
    # ---- head: flatten (channel-major) ++ raw emb -> Linear -> Mish -> Linear; forward + backward in one op ----
    lin1, lin2 = clf.final_block[0], clf.final_block[2]
    fc = cur.chans * cur.length
    if lin1.in_features != fc + clf.model_dim:
        raise ValueError("classifier head does not match the flattened feature size")
    head_off = b.emb_slot(lin1.out_features)
    g = b.act(cur.length, cur.chans)                       # gradient w.r.t. the last downsample's output
    b.head(cur, g, lin1.weight.detach()[:, :fc].reshape(lin1.out_features, cur.chans, cur.length), head_off, lin2.weight.detach(),
           lin2.bias.detach() if lin2.bias is not None else None)
    if forward_only:
        return blocks, (lin1, head_off)

    # ---- backward: g = gradient w.r.t. the output of the tape entry on top ----
    # what lies BELOW an entry decides the epilogue of the op that completes the gradient w.r.t. that entry's input: another
    # block's output (a residual sum: keep the plain gradient for the skip path AND push it through that block's conv2 norm),
    # a downsample's output or the network input (plain)
    def lower_entry(idx: int):
        return tape[idx - 1] if idx > 0 else None

    def finish(idx: int, make_op):
        """Emit the op that completes the gradient w.r.t. tape[idx]'s input; returns (plain gradient slot, pre-norm gradient slot)."""
        below = lower_entry(idx)
        src_act = tape[idx][3] if tape[idx][0] == "block" else tape[idx][2]
        if below is not None and below[0] == "block":
            _, rb_b, _, _, _, s2_b, _ = below
            plain, gu = b.act(src_act.length, src_act.chans), b.act(src_act.length, src_act.chans)
            make_op(gu, dict(gn=rb_b.conv2[1], save=s2_b[0], stats=s2_b[1], dst2=plain))
            return plain, gu
        plain = grad if below is None else b.act(src_act.length, src_act.chans)
        make_op(plain, None)
        return plain, None

    g_plain, g_u2 = g, None
    for idx in range(len(tape) - 1, -1, -1):
        ent = tape[idx]
        if ent[0] == "down":
            _, dn, src = ent
            wt = dn.conv.weight.detach().permute(1, 2, 0)                # [C_in][tap][C_out]
            # y[m] = sum_k W_k x[2m - 1 + k]  =>  gx[2j] = W_1^T gy[j];  gx[2j+1] = W_2^T gy[j] + W_0^T gy[j+1]
            phases = [(wt[:, [1], :], 0, 0), (wt[:, [2, 0], :], 0, 1)]
            gsrc = g_plain
            g_plain, g_u2 = finish(idx, lambda dst, bw: b.conv([gsrc], dst, wt, None, phases=phases, bwd=bw))
            continue
        _, rb, ksz, src, s1, s2, ident = ent
        assert g_u2 is not None, "a block's output gradient always arrives with its conv2 norm already differentiated"
        gu1 = b.act(src.length, rb.conv1[0].out_channels)
        b.conv([g_u2], gu1, _dgrad_eff(rb.conv2[0]), None, pad=ksz // 2,
               bwd=dict(gn=rb.conv1[1], save=s1[0], stats=s1[1], dst2=None))
        gout = g_plain
        wrt = None if ident else rb.residual_conv.weight.detach().permute(1, 2, 0)          # W_r^T as [C_in][1][C_out]
        if ident:
            g_plain, g_u2 = finish(idx, lambda dst, bw: b.conv([gu1], dst, _dgrad_eff(rb.conv1[0]), None, pad=ksz // 2, res=gout, bwd=bw))
        else:
            # gradient w.r.t. the block input = conv1^T g_u1 + W_r^T g_out: two convs whose partial tiles are simply summed -- in one
            # op when that pays, else the skip part first into a slot the main op adds
            def make(dst, bw):
                if FUSE_SKIP and b.conv([gu1], dst, _dgrad_eff(rb.conv1[0]), None, pad=ksz // 2, bwd=bw,
                                        extra=[dict(srcs=[gout], w_eff=wrt, pad=0, bias=None, post=False)]):
                    return
                skip = b.act(src.length, src.chans)
                b.conv([gout], skip, wrt, None)                                             # W_r^T g_out
                b.conv([gu1], dst, _dgrad_eff(rb.conv1[0]), None, pad=ksz // 2, res=skip, bwd=bw)
            g_plain, g_u2 = finish(idx, make)
    return blocks, (lin1, head_off)


def compile_classifier2(clf, horizon: int, max_lds_bytes: int = 160 * 1024, nw: int = NW2_MAX) -> Program2:
    """A HalfJannerUNet1d classifier's forward as its own program (reference nn_classifier/half_jannerunet.py:102-125): the forward
    ops of `_lower_half_janner_grad` and the head, evaluated by the log_p pass of ``cdx_unet2_run`` (``n_steps == 0`` with
    ``logp_out``) -- ``CumRewClassifier.logp`` of a batch with per-sample timesteps in one launch."""
    dev = next(clf.parameters()).device
    b = _Builder2(dev, nw)
    d = clf.in_dim
    x = b.act(horizon, d, persistent=True)
    blocks, (lin1, head_off) = _lower_half_janner_grad(b, clf, horizon, x, None, forward_only=True)
    fcw = lin1.in_features - clf.model_dim
    emb = _emb_table_spec(b, clf, blocks, dev, raw_rows=(lin1.weight.detach()[:, fcw:], lin1.bias.detach(), head_off))
    out = b.op_acts[b.head_index]["dst"]                   # (the head's gradient slot: never written by the log_p pass)
    prog = _finalize2(b, [emb], x, out, horizon, d, clf.emb_dim, max_lds_bytes, [])
    prog.meta["cls_first"], prog.meta["head_op"] = 0, b.head_index
    return prog


def _finalize2(b: "_Builder2", nets_emb: List[dict], x: Act, pred: Act, horizon: int, d: int, emb_dim: int, max_lds_bytes: int,
               persistent: List[Act], grad: Optional[Act] = None, compact: bool = False) -> Program2:
    """LDS plan of one trajectory: [x | persistent slots | prev | stats | stage | arena]; item tables; blob.
    `compact`: the authoritative state x_t and the multistep memory live in GLOBAL memory (the launch's x_out / workspace); the
    LDS state slot is an arena slot that only has to exist from the solver step to the first op of the next forward."""
    nw = b.nw
    off = 0
    if compact:
        x.persistent = False
        b.wrap_live.append(x)
        if b.op_acts[0]["srcs"][0] is not x:
            raise ValueError("compact programs expect the state slot to be read by op 0 only")
    else:
        x.off, off = off, off + x.floats
    for a in persistent:
        a.off, off = off, off + a.floats
    prev_off = -4
    if not compact:
        prev_off, off = off, off + (horizon * d + 3) // 4 * 4
    stats_off, off = off, off + (b.n_stats + 3) // 4 * 4
    stage_off, off = off, off + (b.stage + 3) // 4 * 4
    top = (b.plan_arena(off) + 3) // 4 * 4
    if top * 4 > max_lds_bytes:
        raise ValueError(f"LDS plan needs {top * 4} B > {max_lds_bytes} B per trajectory")
    for op in b.ops:
        op[W2_STATS] += stats_off
    tail, cursor = [], len(b.ops) * op_words(nw)
    for op, items in zip(b.ops, b.op_items):
        for j, rec in enumerate(items[:nw]):              # a wave's first item: fixed offset inside the descriptor
            op[W2_ITEM0 + j * ITEM2_WORDS: W2_ITEM0 + (j + 1) * ITEM2_WORDS] = rec
        op[W2_ITEMS] = cursor
        tail += [w for rec in items[nw:] for w in rec]
        cursor += len(items[nw:]) * ITEM2_WORDS
    ops = np.asarray(b.ops, dtype=np.int32)
    ops_buffer = np.concatenate([ops.reshape(-1), np.asarray(tail, dtype=np.int64).astype(np.int32)])
    b.add(torch.zeros(RING2 * 256))          # the ring prefetch reads a ring of records from an item's first one, whatever its length
    blob = torch.cat(b.chunks).contiguous()
    return Program2(ops=ops, ops_buffer=ops_buffer, blob=blob, traj_floats=top, x_off=x.data_off,
                    x_stride=x.stride, pred_off=pred.data_off, pred_stride=pred.stride, prev_off=prev_off, stage_off=stage_off,
                    horizon=horizon, dim=d, emb_dim=emb_dim, n_emb=b.n_emb, embtab=nets_emb[0], macs_per_forward=b.macs,
                    n_conv=len(b.ops), meta={"blob_floats": b.blob_len, "stats_off": stats_off}, nw=nw,
                    embtabs=nets_emb, grad_off=-1 if grad is None else grad.data_off, grad_stride=0 if grad is None else grad.stride,
                    compact=compact, ws_floats=((horizon * d + 3) // 4 * 4) if compact else 0)


def compile_janner2(net, horizon: int, max_lds_bytes: int = 160 * 1024, allow_4x4: bool = True, nw: int = NW2, compact: bool = False,
                    max_stage: Optional[int] = None, member: Tuple[int, int] = (0, 1), grouped: bool = False,
                    alias_residual: Optional[bool] = None) -> Program2:
    """Lower a JannerUNet1d (reference nn_diffusion/jannerunet.py:98-201) for `horizon` positions and `nw` waves per workgroup.
    `compact`: the small-LDS variant for THREE trajectories per workgroup (in-place residual outputs, capped staging area)."""
    why = supports_janner(net)
    if why is not None:
        raise ValueError(why)
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    b.allow_4x4 = allow_4x4
    b.alias_residual = compact if alias_residual is None else alias_residual
    b.member = member
    b.grouped = grouped
    if member[1] > 1:
        if compact or nw != NW2_MAX:
            raise ValueError("split / grouped programs: the 8-wave, state-in-LDS form only")
        b.fuse_max = 1 << 30                  # (a member's K slices are short: the 1x1 skips always ride in their block's second conv)
    if max_stage is not None:
        b.max_stage = max_stage
    d = net.in_dim
    x = b.act(horizon, d, persistent=True)
    t, fc, blocks = _lower_janner(b, net, horizon, x)
    pred = b.act(horizon, d)                 # arena slot: written by the last op, read by the solver step right after it
    b.conv([t], pred, _conv1d_eff(fc[3]), fc[3].bias, pred=True)
    emb = _emb_table_spec(b, net, blocks, dev)
    prog = _finalize2(b, [emb], x, pred, horizon, d, net.emb_dim, max_lds_bytes, [], compact=compact)
    prog.meta["xchg_floats"] = b.xchg_floats
    prog.meta["n_gops"] = sum(1 for oa in b.op_acts if oa.get("gop"))
    return prog


def compile_janner2_split(net, horizon: int, k: int, max_lds_bytes: int = 160 * 1024) -> Program2:
    """One trajectory over k workgroups of an XCD (small batches: B x k <= 256 workgroups).  Every member holds the whole activation
    set in its own LDS and runs the whole op list, but computes only its 1/k of the row tiles (and GroupNorm groups) of the ops that
    can be cut that way; after such an op the members all-gather the op's output through a 4 KB tile in global memory (same L2: no
    agent-scope fence, tools/xwg_exchange_probe.hip).  The k member views share the blob, the LDS plan and the op count; their
    descriptors are laid out [member 0 ops | member 1 ops | ... | item tails], so member m's op i is descriptor m * n_ops + i."""
    if k not in (2, 4):
        raise ValueError("split factor 2 or 4")
    members = [compile_janner2(net, horizon, max_lds_bytes=max_lds_bytes, nw=NW2_MAX, member=(m, k)) for m in range(k)]
    return _merge_members(members, k)


def _merge_members(members: List[Program2], k: int) -> Program2:
    """One program out of the k member views (same blob, LDS plan and op count): descriptors [member 0 | member 1 | ... | item tails]."""
    p0 = members[0]
    n_ops, opw = len(p0.ops), op_words(NW2_MAX)
    for p in members[1:]:
        same = (len(p.ops), p.traj_floats, p.x_off, p.pred_off, p.prev_off, p.stage_off, p.n_emb, p.blob.numel()) == \
               (n_ops, p0.traj_floats, p0.x_off, p0.pred_off, p0.prev_off, p0.stage_off, p0.n_emb, p0.blob.numel())
        if not same or not torch.equal(p.blob, p0.blob):
            raise ValueError("member views of a split program disagree on the LDS plan or the blob")
    ops_all, tails, base = [], [], k * n_ops * opw
    for p in members:
        ops = p.ops.copy()
        tail = p.ops_buffer[n_ops * opw:]
        ops[:, W2_ITEMS] = ops[:, W2_ITEMS] - n_ops * opw + base        # tail cursors: re-based behind ALL members' descriptors
        base += tail.size
        ops_all.append(ops)
        tails.append(tail)
    p0.meta["split_k"] = k
    p0.meta["member_ops"] = ops_all                      # (tail cursors re-based: what oracle/lane_sim2.py interprets per member)
    p0.ops = ops_all[0]
    p0.ops_buffer = np.concatenate([np.concatenate(ops_all).reshape(-1)] + tails).astype(np.int32)
    p0.meta["xchg_floats"] = max(p.meta["xchg_floats"] for p in members)
    if p0.meta["xchg_floats"] == 0:
        raise ValueError("split program: no op of this net streams enough weights to be cut over the members")
    return p0


def compile_janner2_group(net, horizon: int, k: int, max_lds_bytes: int = 160 * 1024) -> Program2:
    """k trajectories over the k workgroups of a group on one XCD -- the full-batch counterpart of `compile_janner2_split` (B = 256 on
    256 CUs: no idle CU to give a trajectory).  Member m owns trajectory m of its group and runs the whole op list on it, EXCEPT the ops
    that stream enough weights to be bound by the L2 -> CU path (config 2: the 0.6-1.3 MB layers at 4 positions): those are GROUPED --
    the member computes its 1/k of the output channels (whole GroupNorm lane groups) for ALL k trajectories, whose k x 4 positions fill
    the 16 columns of a 16x16x4 tile (instead of 4 of a 4x4x1 block), so every streamed weight record feeds k trajectories and each CU
    streams 1/k of the layer.  The slots such ops touch hold the group's k trajectories side by side (k sub-slots); after a grouped op the
    members all-gather its output through the group's tile in L2 (the exchange of the split programs), and an ordinary op that feeds a
    grouped one publishes its whole trajectory the same way.  Descriptor layout as for split programs: [member 0 ops | member 1 ops | ...
    | item tails]."""
    if k not in (2, 4):
        raise ValueError("group size 2 or 4")
    members = None
    for alias in (False, True):                       # in-place residual outputs only if the plain plan does not fit
        try:
            members = [compile_janner2(net, horizon, max_lds_bytes=max_lds_bytes, nw=NW2_MAX, member=(m, k), grouped=True,
                                       alias_residual=alias) for m in range(k)]
            break
        except ValueError:
            if alias:
                raise
    p0 = _merge_members(members, k)
    p0.meta["group_k"] = k
    del p0.meta["split_k"]
    if p0.meta["n_gops"] == 0:
        raise ValueError("grouped program: no op of this net streams enough weights to be grouped")
    return p0


def compile_guided2(net, clf, horizon: int, max_lds_bytes: int = 160 * 1024, nw: int = NW2_MAX, save_global: bool = False,
                    max_stage: Optional[int] = None, compact: bool = False, member: Tuple[int, int] = (0, 1), grouped: bool = False,
                    alias_residual: Optional[bool] = None) -> Program2:
    """Denoiser forward + classifier forward/backward as ONE op list (classifier-guided sampling, reference
    diffusionsde.py:153-173): ops [0, n_den) write the prediction, the rest writes d log p / d x_t into the gradient slot; the
    kernel's solver step shifts the prediction by cg_scale[step] * gradient before clipping.  Both networks read the state slot."""
    why = supports_janner(net)
    if why is not None:
        raise ValueError(why)
    if clf.in_dim != net.in_dim:
        raise ValueError("classifier and denoiser disagree on the state dimension")
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    b.save_global = save_global          # saved x_hat tensors in global memory: the LDS plan shrinks enough for two trajectories
    if max_stage is not None:
        b.max_stage = max_stage
    b.alias_residual = compact if alias_residual is None else alias_residual
    b.member, b.grouped = member, grouped
    if member[1] > 1:                    # member view of a GROUPED / SPLIT guided program (compile_guided2_group / _split)
        if compact or nw != NW2_MAX:
            raise ValueError("grouped / split guided programs: the 8-wave, state-in-LDS form only")
        b.fuse_max = 1 << 30
    d = net.in_dim
    if compact:
        b.ws_floats = (horizon * d + 3) // 4 * 4         # workspace of a trajectory: [multistep memory | saved tensors]
    x = b.act(horizon, d, persistent=True)
    # prediction and gradient are arena slots that stay live until the solver step has read them (the prediction through all of the
    # classifier's ops, whose own footprint is small next to the denoiser's peak): ~2 slots less than keeping them persistent
    pred, grad = b.act(horizon, d), b.act(horizon, d)
    b.keep_to_end += [pred, grad]
    t, fc, blocks = _lower_janner(b, net, horizon, x)
    b.conv([t], pred, _conv1d_eff(fc[3]), fc[3].bias, pred=True)
    n_den = len(b.ops)
    b.group_on = False                   # (grouped guided programs: the classifier's layers stay on the member's own trajectory)
    emb_den = _emb_table_spec(b, net, blocks, dev)
    xc = x
    if compact:
        # `compact` (the largest nets that fit at all: model_dim 64 at H = 64): the state and the multistep memory live in global
        # memory, the denoiser's LDS copy of x_t dies with its first op, and the classifier gets its own copy through a load op
        xc = b.act(horizon, d)
        b.load_state(xc)
    cblocks, (lin1, head_off) = _lower_half_janner_grad(b, clf, horizon, xc, grad)
    fcw = lin1.in_features - clf.model_dim
    emb_clf = _emb_table_spec(b, clf, cblocks, dev, raw_rows=(lin1.weight.detach()[:, fcw:], lin1.bias.detach(), head_off))
    prog = _finalize2(b, [emb_den, emb_clf], x, pred, horizon, d, net.emb_dim, max_lds_bytes, [], grad=grad, compact=compact)
    prog.meta["n_den"] = n_den
    prog.meta["cls_first"], prog.meta["head_op"] = n_den, b.head_index      # final log_p forward: ops [cls_first, head_op] once more
    prog.ws_floats = b.ws_floats
    prog.meta["xchg_floats"] = b.xchg_floats
    prog.meta["n_gops"] = sum(1 for oa in b.op_acts if oa.get("gop"))
    return prog


def compile_guided2_split(net, clf, horizon: int, k: int, max_lds_bytes: int = 160 * 1024) -> Program2:
    """The guided program for SMALL batches (B x k <= 256 workgroups): one trajectory over k workgroups of an XCD as in
    `compile_janner2_split` -- the DENOISER's ops that can be cut by row tiles are, with an all-gather behind each; the classifier's
    forward / backward ops, the solver step and the final log_p pass are computed by every member on its own copy of the trajectory."""
    if k not in (2, 4):
        raise ValueError("split factor 2 or 4")
    members = [compile_guided2(net, clf, horizon, max_lds_bytes=max_lds_bytes, member=(m, k)) for m in range(k)]
    return _merge_members(members, k)


def compile_guided2_group(net, clf, horizon: int, k: int, max_lds_bytes: int = 160 * 1024, save_global: bool = False) -> Program2:
    """The guided program with its DENOISER's stream-bound layers grouped (`compile_janner2_group`): k trajectories over the k workgroups
    of a group, member m computes 1/k of the output channels of those layers for all k trajectories and the members all-gather through
    L2; every other op -- the rest of the denoiser, the classifier's forward and backward ops with their saved tensors, the solver
    step, the final log_p pass -- runs on the member's own trajectory as in the ordinary guided program (`cdx_unet2_kernel<1, 8, true,
    ..., split>`)."""
    if k not in (2, 4):
        raise ValueError("group size 2 or 4")
    members = None
    for alias in (False, True):
        try:
            members = [compile_guided2(net, clf, horizon, max_lds_bytes=max_lds_bytes, member=(m, k), grouped=True, alias_residual=alias,
                                       save_global=save_global) for m in range(k)]
            break
        except ValueError:
            if alias:
                raise
    p0 = _merge_members(members, k)
    p0.meta["group_k"] = k
    del p0.meta["split_k"]
    if p0.meta["n_gops"] == 0:
        raise ValueError("grouped guided program: no op of the denoiser streams enough weights to be grouped")
    return p0
