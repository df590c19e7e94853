"""Network -> device program for the second-generation fused U-Net kernel (``csrc/cdx_unet2.hip``).

Same idea as ``program.py`` (one launch = the whole ``sample()`` loop, activations resident in LDS, weights streamed as
1-KiB MFMA records).  The format is shaped by what the round-2 profiles showed: with one workgroup per CU the kernel is bound by
per-op latency and by *instruction issue* (a wave64 issues about one instruction every four clocks), so everything the K loop
does per weight record beyond "wait, 4 MFMAs, one LDS read, one global load" was designed out on the host side:

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
Activations: channel-last rows ``slot[(pos + HALO2) * stride + c]``, ``stride = pad16(C) + 4``.  Conv records: identical lane
layouts to program.py (MODE_16X16 / MODE_4X4).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .program import (GN_EPS, MODE_16X16, MODE_4X4, _conv1d_eff, _convT1d_eff, _fbits, slot_stride, supports_janner)

HDR_WORDS = 32                # 25 descriptor words, padded
ITEM2_WORDS = 8
NW2 = 4                       # waves per workgroup, default shape (one per SIMD)
NW2_MAX = 8                   # ... two per SIMD
RING2 = 16                    # weight records in flight per wave, 4-wave shape
RING2_NW8 = 8                 # ... 8-wave shape
GROUPS2 = 8                   # GroupNorm groups the epilogue partition is built for (32 lanes per group)
MAX_NK2 = 4                   # float4 items a lane may own in the epilogue
HALO2 = 2                     # zero rows on either side of a slot: covers kernel sizes <= 5 (pad <= 2)

(W2_KIND, W2_FLAGS, W2_COUT, W2_LOUT, W2_LCOLS, W2_CSTRIDE, W2_OSTRIDE, W2_MODE, W2_NT, W2_NITEMS, W2_ITEMS, W2_DST,
 W2_DST_STRIDE, W2_SSTRIDE, W2_KSPLIT, W2_BOFF, W2_GAMMA, W2_BETA, W2_EMB, W2_RES, W2_RES_STRIDE, W2_CG4_SHIFT, W2_INV_CNT,
 W2_NK, W2_COUTP) = range(25)
W2_ITEM0 = 32                 # items 0..nw-1 inline; items nw.. in the tail table at W2_ITEMS


def op_words(nw: int) -> int:
    """int32 words per op descriptor of a program compiled for `nw` waves (cdx.h: CDX2_OP_WORDS(n_waves))."""
    return HDR_WORDS + ITEM2_WORDS * nw


def ring_depth(nw: int) -> int:
    return {4: RING2, 8: RING2_NW8}[nw]
# item record: blob offset of the first record, record count, start cursor (tap | chunk << 8), stage offset of the partial tile,
# first column, (pad | output-position offset << 8), source slot (float offset | row stride << 16), chunks per tap
I2_WOFF, I2_NQ, I2_TAPCC, I2_PART, I2_COL0, I2_PADOOFF, I2_SRCSTR, I2_CCN = range(8)

F2_GN, F2_EMB, F2_RES, F2_PRED = 1, 2, 4, 8


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

    @property
    def stride(self):
        return slot_stride(self.chans)

    @property
    def floats(self):
        return (self.length + 2 * HALO2) * self.stride

    @property
    def data_off(self):               # float offset of position 0
        return self.off + HALO2 * self.stride


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

    def lds_bytes(self, traj_per_wg: int) -> int:
        return 4 * self.traj_floats * traj_per_wg


def _records(w_eff: torch.Tensor, mode: int) -> Tuple[torch.Tensor, int]:
    """w_eff [C_out][taps][C_in] -> records [n_row_tiles][taps * cc][64][4] and cc = chunks per tap.
    Lane layouts as program.py:pack_conv: MODE_16X16 lane = k4*16 + row, 16 K per record; MODE_4X4 lane = row, 4 K."""
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
        self.op_acts: List[Tuple[List[Act], Act]] = []     # (slots read: sources [+ residual], slot written) per op
        self.op_item_src: List[List[int]] = []            # per op, per item: index of the source slot it reads
        self.op_items: List[list] = []
        self.chunks: List[torch.Tensor] = []
        self.blob_len = 0
        self.acts: List[Act] = []
        self.stage = 0
        self.macs = 0
        self.n_emb = 0
        self.allow_4x4 = True

    def add(self, t: torch.Tensor, pad_to: int = 4) -> int:
        t = t.detach().to(device=self.device, dtype=torch.float32).reshape(-1)
        off = self.blob_len
        pad = (-t.numel()) % pad_to
        if pad:
            t = torch.cat([t, torch.zeros(pad, device=self.device)])
        self.chunks.append(t)
        self.blob_len += t.numel()
        return off

    def act(self, length: int, chans: int, persistent=False) -> Act:
        a = Act(length, chans, len(self.acts), persistent=persistent)
        self.acts.append(a)
        return a

    def emb_slot(self, c_out: int) -> int:
        off = self.n_emb
        self.n_emb += pad32(c_out)
        return off

    def conv(self, srcs: List[Act], dst: Act, w_eff: torch.Tensor, bias: torch.Tensor, *, stride=1, pad=0, transposed=False,
             gn: Optional[nn.Module] = None, emb_off: int = -1, res: Optional[Act] = None, pred: bool = False):
        """One fused op: conv over the channel concat of `srcs` -> [GroupNorm -> Mish] -> [+ emb] -> [+ residual slot `res` (may be
        `dst`: accumulate)] -> dst.  `transposed`: ConvTranspose1d(k=4, stride=2, pad=1) as two 2-tap convs, one per output parity."""
        c_out, taps, c_in = w_eff.shape
        l_out = dst.length
        assert c_in == sum(a.chans for a in srcs) and dst.chans == c_out and 1 <= len(srcs) <= 2
        if transposed:
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
        mode = MODE_4X4 if (l_cols <= 8 and c_out % 64 == 0 and self.allow_4x4) else MODE_16X16
        rows, cols = (16, 16) if mode == MODE_16X16 else (64, 4)
        nt = 2 if (mode == MODE_4X4 and l_cols > 4) else 1
        n_rt = -(-c_out // rows)
        n_cg = -(-l_cols // (nt * cols))
        tiles = n_rt * n_cg * len(phases)
        coutp = pad32(c_out)
        sstride = coutp + 4
        # record stream of a row tile: [source 0: taps x chunks | source 1: taps x chunks]; a K slice never straddles the sources
        streams = []
        for w, _, _ in phases:
            segs, lo = [], 0
            for a in srcs:
                segs.append(_records(w[:, :, lo:lo + a.chans], mode))
                lo += a.chans
            streams.append(segs)
        seg_n = [r.shape[1] for r, _ in streams[0]]
        nqt = sum(seg_n)
        # K slices: cut every source's record range evenly; with two sources (a concat) the slices never straddle the boundary,
        # whatever the tile count -- each source gets at least one slice and the epilogue sums the staged partials
        nw, ring = self.nw, ring_depth(self.nw)
        per_src = max(1, (nw // tiles if tiles < nw else 1) // len(srcs))
        per_src = [min(per_src, n) for n in seg_n]
        ksplit = sum(per_src)
        cuts, base = [], 0                                  # (first record, one past the last, source index) per slice
        for si, (n, k) in enumerate(zip(seg_n, per_src)):
            # long streams are cut on multiples of the ring depth: the kernel's immediate-offset steady loop needs a slice to start
            # on a ring-aligned chunk of its tap
            al = ring if n >= 2 * ring * k else 1
            edge = [min(n, (j * n // k + al // 2) // al * al) for j in range(k)] + [n]
            cuts += [(base + edge[j], base + edge[j + 1], si, base) for j in range(k)]
            base += n
        woffs = [self.add(torch.cat([r for r, _ in segs], dim=1).contiguous()) for segs in streams]
        items, item_src = [], []
        for item in range(tiles * ksplit):
            tile, ks = item % tiles, item // tiles
            ph, tile = tile % len(phases), tile // len(phases)
            rt, cgi = tile % n_rt, tile // n_rt
            q0, q1, si, sbase = cuts[ks]
            ccn = streams[ph][si][1]
            items.append([woffs[ph] + (rt * nqt + q0) * 256, q1 - q0, ((q0 - sbase) // ccn) | (((q0 - sbase) % ccn) << 8),
                          ks * l_out * sstride + rt * rows, cgi * nt * cols, phases[ph][1] | (phases[ph][2] << 8), 0, ccn])
            item_src.append(si)
        words = {W2_KIND: 0, W2_COUT: c_out, W2_LOUT: l_out, W2_LCOLS: l_cols, W2_CSTRIDE: cstride, W2_OSTRIDE: ostride,
                 W2_MODE: mode, W2_NT: nt, W2_NITEMS: len(items), W2_DST_STRIDE: dst.stride, W2_SSTRIDE: sstride,
                 W2_KSPLIT: ksplit, W2_BOFF: self.add(_padded(bias, coutp)), W2_COUTP: coutp}
        cg = coutp // GROUPS2
        cg4 = cg // 4
        assert cg4 & (cg4 - 1) == 0 and cg4 <= 32, f"C_out {c_out}: channels per group / 4 must be a power of two <= 32"
        nk = -(-(cg4 * l_out) // 32)
        if nk > MAX_NK2:
            raise ValueError(f"epilogue: {cg4 * l_out} float4 items per group > {32 * MAX_NK2} (horizon too long for v2)")
        words[W2_CG4_SHIFT], words[W2_NK] = cg4.bit_length() - 1, nk
        flags = 0
        if gn is not None:
            # the epilogue cuts pad32(C_out) channels into 8 lane groups; a real group must be exactly one of them (C_out = 16 with
            # 4 groups of 4 is fine: lane groups 4..7 then work on pad channels that are never stored)
            if c_out % gn.num_groups or c_out // gn.num_groups != cg or abs(gn.eps - GN_EPS) > 1e-12:
                raise ValueError(f"v2 epilogue needs GroupNorm groups of pad32(C)/8 channels (C={c_out}, G={gn.num_groups})")
            flags |= F2_GN
            words[W2_INV_CNT] = _fbits(1.0 / (cg * l_out))
            words[W2_GAMMA], words[W2_BETA] = self.add(_padded(gn.weight, coutp)), self.add(_padded(gn.bias, coutp))
        if emb_off >= 0:
            flags |= F2_EMB
            words[W2_EMB] = emb_off
        if res is not None:
            assert res.chans == c_out and res.length == l_out
            flags |= F2_RES
            words[W2_RES_STRIDE] = res.stride
        if pred:
            flags |= F2_PRED
        words[W2_FLAGS] = flags
        op = [0] * op_words(self.nw)
        for k, v in words.items():
            op[k] = int(v)
        self.ops.append(op)
        self.op_acts.append((list(srcs) + ([res] if res is not None else []), dst))
        self.op_items.append(items)
        self.op_item_src.append(item_src)
        self.stage = max(self.stage, ksplit * l_out * sstride)
        self.macs += c_out * l_out * taps * c_in // (2 if transposed else 1)

    def plan_arena(self, base: int) -> int:
        """First-fit interval allocation over op liveness (same policy as program.py); patches slot offsets into the ops."""
        first, last = {}, {}
        for i, (reads, writes) in enumerate(self.op_acts):
            for a in reads + [writes]:
                first.setdefault(a.uid, i)
                last[a.uid] = i
        live: List[Act] = []
        top = base
        for a in sorted((a for a in self.acts if not a.persistent and a.uid in first), key=lambda a: first[a.uid]):
            t = first[a.uid]
            live = [b for b in live if last[b.uid] >= t]
            pos = base
            for lo, hi in sorted((b.off, b.off + b.floats) for b in live):
                if lo - pos >= a.floats:
                    break
                pos = max(pos, hi)
            a.off = pos
            live.append(a)
            top = max(top, pos + a.floats)
        for op, (reads, dst), items, item_src in zip(self.ops, self.op_acts, self.op_items, self.op_item_src):
            op[W2_DST] = dst.off                                       # slot base = first halo row
            if op[W2_FLAGS] & F2_RES:
                op[W2_RES] = reads[-1].off
            for rec, si in zip(items, item_src):
                assert reads[si].off < (1 << 16) and reads[si].stride < (1 << 15)
                rec[I2_SRCSTR] = reads[si].off | (reads[si].stride << 16)
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


def compile_janner2(net, horizon: int, max_lds_bytes: int = 160 * 1024, allow_4x4: bool = True, nw: int = NW2) -> Program2:
    """Lower a JannerUNet1d (reference nn_diffusion/jannerunet.py:98-201) for `horizon` positions and `nw` waves per workgroup."""
    why = supports_janner(net)
    if why is not None:
        raise ValueError(why)
    dev = next(net.parameters()).device
    b = _Builder2(dev, nw)
    b.allow_4x4 = allow_4x4
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
            b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1])
            b.conv(srcs, out, _conv1d_eff(rb.residual_conv), rb.residual_conv.bias, res=out)     # out += W_r x + b_r
        return out

    x = b.act(horizon, d, persistent=True)
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
    t = b.act(horizon, md)
    b.conv([cur], t, _conv1d_eff(fc[0]), fc[0].bias, pad=2, gn=fc[1])
    pred = b.act(horizon, d)                 # arena slot: written by the last op, read by the solver step right after it
    b.conv([t], pred, _conv1d_eff(fc[3]), fc[3].bias, pred=True)

    # embedding MLP (evaluated by cdx_unet2_embtab, once per schedule): transposed [n_in][n_out] weights
    emb = {"emb_dim": net.emb_dim, "hidden": net.map_emb[0].out_features, "md": net.map_emb[2].out_features,
           "n_emb": b.n_emb}
    emb["w0"], emb["b0"] = b.add(net.map_emb[0].weight.t().contiguous()), b.add(net.map_emb[0].bias)
    emb["w2"], emb["b2"] = b.add(net.map_emb[2].weight.t().contiguous()), b.add(net.map_emb[2].bias)
    w_all = torch.zeros(b.n_emb, emb["md"], device=dev)
    b_all = torch.zeros(b.n_emb, device=dev)
    for rb, off in blocks:
        lin = rb.emb_mlp[1]
        w_all[off:off + lin.out_features] = lin.weight.detach().to(dev)
        b_all[off:off + lin.out_features] = lin.bias.detach().to(dev)
    emb["w3"], emb["b3"] = b.add(w_all.t().contiguous()), b.add(b_all)
    b.macs += net.emb_dim * emb["hidden"] + emb["hidden"] * emb["md"] + emb["md"] * sum(rb.emb_mlp[1].out_features for rb, _ in blocks)

    # ---- LDS plan of one trajectory: [x | prev | stage | arena] ----
    off = 0
    x.off, off = off, off + x.floats
    prev_off, off = off, off + (horizon * d + 3) // 4 * 4
    stage_off, off = off, off + (b.stage + 3) // 4 * 4
    top = (b.plan_arena(off) + 3) // 4 * 4
    if top * 4 > max_lds_bytes:
        raise ValueError(f"LDS plan needs {top * 4} B > {max_lds_bytes} B per trajectory")
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
                    horizon=horizon, dim=d, emb_dim=net.emb_dim, n_emb=b.n_emb, embtab=emb, macs_per_forward=b.macs,
                    n_conv=len(b.ops), meta={"blob_floats": b.blob_len}, nw=nw)
