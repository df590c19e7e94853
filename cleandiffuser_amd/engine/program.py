"""Network -> device-program compiler for the fused gfx950 executor.

A backbone (``nn.Module``) is lowered ONCE per weight version into

* ``ops``   int32 [n_ops, OP_WORDS]   -- a flat list of layer descriptors the kernel interprets in order
* ``blob``  float32 [n]               -- every parameter, pre-packed in the exact order the kernel streams it
* an LDS plan (float offsets)         -- where every activation lives; nothing is ever written back to HBM

Data layout contract with ``csrc/cdx_unet1d.hip`` (one workgroup = one trajectory, 8 waves):

Activations (LDS):  channel-last rows with a 2-row zero halo:  ``slot[(pos + 2) * stride + c]``,
    ``stride = pad16(C) + 4`` floats (the +4 keeps ``ds_read_b128`` of 16 different rows on different banks).
Conv weights (HBM, streamed through L2/MALL): implicit-GEMM ``out[co][n] = sum_K W[co][K] X[K][n]`` tiled for
    ``v_mfma_f32_16x16x4_f32``; K is enumerated as (source, tap, 16-channel chunk); per (16-row tile ct, chunk q)
    the blob holds 64 lanes x float4 = 1 KiB contiguous so a wave fetches it with ONE ``global_load_dwordx4``:
        ``packed[ct][q][lane][m] = W[ct*16 + (lane & 15)][src, tap, cc*16 + 4*(lane >> 4) + m]``
    (lane>>4 is the MFMA k index; the 4 floats m feed 4 consecutive MFMAs; A and B use the same K permutation).
Linear weights: transposed ``[n_in][n_out]`` so consecutive lanes read consecutive floats.

Narrow layers (<= 8 positions, C_out % 64 == 0) use MODE_4X4 instead: ``v_mfma_f32_4x4x1_16b_f32`` computes 16 independent
    4x4 outer products, i.e. 64 output channels x 4 positions per instruction with no padded columns; the record is
        ``packed[ct64][q][lane][m] = W[ct64*64 + lane][src, tap, cc*4 + m]``   (one K value per MFMA, 4 per record).
All integer divisions the kernel would need (work-item decode, K-range boundaries, cursor start) are done here and
shipped as an item table behind the ops (``I_*`` words), because scalar division costs ~200 cycles on the device.

Op words: see ``W_*`` constants below; flags ``F_*``.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

OP_WORDS = 40
ITEM_WORDS = 8
OP_LOAD_TEMB, OP_LINEAR, OP_CONV, OP_FLATTEN, OP_FILL, OP_LOAD_COND = 0, 1, 2, 3, 4, 5
MODE_16X16, MODE_4X4 = 0, 1        # MFMA shape of a conv: 16x16x4 (16 rows x 16 cols x 16 K per record) or
                                   # 4x4x1 x16 blocks (64 rows x 4 cols x 4 K per record) for <= 8 positions

# ---- word indices (all ops) ---------------------------------------------------------------------- #
W_KIND = 0
# conv
(W_COUT, W_COUT16, W_LOUT, W_TAPS, W_CSTRIDE, W_CPAD, W_TRANSPOSED,
 W_SRCA, W_SRCA_STRIDE, W_CA_CHUNKS, W_SRCB, W_SRCB_STRIDE, W_CB_CHUNKS,
 W_DST, W_DST_STRIDE, W_DST_ROWS, W_WOFF, W_BOFF, W_FLAGS, W_GROUPS, W_GAMMA, W_BETA,
 W_EMB, W_RES, W_RES_STRIDE, W_KSPLIT, W_NCHUNKS, W_LIN,
 W_MODE, W_ITEMS, W_NITEMS, W_INV_CNT, W_CG, W_CG_SHIFT, W_INV_COUT,
 W_ACT, W_NORM, W_SCALE, W_DST_COFF) = range(1, 40)
# item record (ITEM_WORDS int32 each, appended to the ops buffer): one K-range of one row tile = one wave's job
I_WOFF, I_PART, I_NQ, I_ONB, I_TAP, I_CC = range(6)
# linear / load_temb (reuse low word indices)
L_NIN, L_NOUT, L_SRC, L_DST, L_WOFF, L_BOFF, L_FLAGS, L_DST2 = range(1, 9)
# flatten (slot -> vector, channel-major like torch .flatten(1) of (b, C, L)): L_NIN = C, L_NOUT = L, L_SRC = slot,
# L_DST = vector, L_WOFF = slot stride

F_GN_MISH, F_ADD_EMB, F_ADD_RES, F_ACCUM, F_DST_PRED, F_POST_MISH, F_RAW_COPY, F_SCALE, F_KEEP_DST, F_FILM = \
    1, 2, 4, 8, 16, 32, 64, 128, 256, 512
# activation ids (W_ACT) and normalisation modes (W_NORM: statistics over the whole slot group / per column)
ACT_NONE, ACT_MISH, ACT_GELU_ERF, ACT_LEAKY, ACT_SILU, ACT_RELU, ACT_GELU_TANH, ACT_MISH_GRAD, ACT_TANH = range(9)
NORM_NONE, NORM_SLOT_GROUP, NORM_COLUMN = 0, 1, 2

HALO = 0          # slots carry no halo rows: out-of-range conv taps read a shared all-zero row (Program.zrow_off)
N_WAVES = 8
GN_EPS = 1e-5


def _fbits(x: float) -> int:
    """fp32 bit pattern as a (signed) int32 op word."""
    return int(np.float32(x).view(np.int32))


def pad16(c: int) -> int:
    return (c + 15) // 16 * 16


def slot_stride(c: int) -> int:
    return pad16(c) + 4


def slot_floats(length: int, c: int) -> int:
    return (length + 2 * HALO) * slot_stride(c)


@dataclass
class Act:
    """An activation tensor (L positions x C channels) living in an LDS slot."""
    length: int
    chans: int
    uid: int
    off: int = -1                     # float offset into the workgroup's LDS, assigned by the allocator
    persistent: bool = False

    @property
    def stride(self):
        return slot_stride(self.chans)

    @property
    def floats(self):
        return slot_floats(self.length, self.chans)


@dataclass
class Program:
    ops: np.ndarray                    # int32 [n_ops, OP_WORDS]
    ops_buffer: np.ndarray             # int32 1-D: the ops followed by the item tables (what the device gets)
    blob: torch.Tensor                 # float32 1-D (device of the module)
    lds_floats: int
    x_off: int
    x_stride: int
    pred_off: int
    pred_stride: int
    pred_branch_floats: int            # distance between the two CFG prediction slots
    prev_off: int                      # dense [H*D] buffer for the multistep solvers
    vec_off: int
    scratch_off: int
    scratch_floats: int
    desc_off: int                      # LDS home of the kernel's copy of ops_buffer
    prof_off: int                      # LDS home of the profiling stamps (u64, so an even float offset)
    horizon: int
    dim: int
    emb_dim: int
    macs_per_forward: int              # algorithmic MACs (conv + linear), for the roofline accounting
    n_conv: int = 0
    desc_words: int = 0                # how many words of ops_buffer the kernel copies into LDS
    items_in_lds: bool = True          # False: only the ops are copied, item tables are read from the global buffer
    out_vec_off: int = 0               # vector-output programs (classifier heads): where the result lives
    out_vec_len: int = 0
    tile: int = 0                      # > 0: batch-tiled MLP program, `horizon` = samples per workgroup
    cond_slot_off: int = 0             # tile programs: per-sample condition features live in this slot ...
    cond_slot_stride: int = 0
    cond_coff: int = 0                 # ... at this channel offset ...
    cond_dim: int = 0                  # ... this many of them per sample
    zero_off: int = 0                  # the persist slots as one contiguous range (what the kernel clears at start)
    zero_floats: int = 0
    zrow_off: int = 0                  # the shared all-zero row (inside the zero range)
    persist_slots: list = field(default_factory=list)   # (offset, floats) of extra kernel-lifetime slots to zero once
    meta: dict = field(default_factory=dict)


class _Builder:
    def __init__(self, device):
        self.device = device
        self.ops: List[List[int]] = []
        self.op_acts: List[Tuple[List[Act], Optional[Act]]] = []   # (reads, writes) per op for liveness
        self.chunks: List[torch.Tensor] = []
        self.blob_len = 0
        self.acts: List[Act] = []
        self.vec_len = 0
        self.scratch = 0
        self.macs = 0
        self.n_conv = 0
        self.op_items: List[Optional[list]] = []           # per op: item records (convs) or None
        self.allow_4x4 = True

    # ---------------- parameter blob ---------------- #
    def add(self, t: torch.Tensor) -> int:
        t = t.detach().to(device=self.device, dtype=torch.float32).reshape(-1)
        off = self.blob_len
        pad = (-t.numel()) % 4                          # keep every record 16-byte aligned
        if pad:
            t = torch.cat([t, torch.zeros(pad, device=self.device)])
        self.chunks.append(t)
        self.blob_len += t.numel()
        return off

    def pack_conv(self, w_eff: torch.Tensor, split: Sequence[int], mode: int) -> Tuple[int, int, List[int]]:
        """w_eff [C_out][taps][C_in_total] (implicit-GEMM view), split = channel count per source.
        Returns (blob offset, n_chunks, chunks-per-tap per source)."""
        c_out, taps, c_in = w_eff.shape
        assert sum(split) == c_in
        rows, kch = (16, 16) if mode == MODE_16X16 else (64, 4)
        n_ct = -(-c_out // rows)
        parts, per_src, lo = [], [], 0
        for cs in split:
            w = w_eff[:, :, lo:lo + cs]
            lo += cs
            csp = -(-cs // kch) * kch
            wp = torch.zeros(n_ct * rows, taps, csp, device=w.device, dtype=torch.float32)
            wp[:c_out, :, :cs] = w
            cc = csp // kch
            if mode == MODE_16X16:
                # [ct, i, tap, cc, k4, m] -> [ct, tap, cc, k4, i, m] -> [ct, tap*cc, 64, 4]   (lane = k4*16 + i)
                wp = wp.reshape(n_ct, 16, taps, cc, 4, 4).permute(0, 2, 3, 4, 1, 5)
            else:
                # [ct, i, tap, cc, m] -> [ct, tap, cc, i, m] -> [ct, tap*cc, 64, 4]           (lane = i)
                wp = wp.reshape(n_ct, 64, taps, cc, 4).permute(0, 2, 3, 1, 4)
            parts.append(wp.reshape(n_ct, taps * cc, 64, 4))
            per_src.append(cc)
        packed = torch.cat(parts, dim=1).contiguous()
        return self.add(packed), packed.shape[1], per_src

    # ---------------- LDS objects ---------------- #
    def act(self, length: int, chans: int, persistent=False) -> Act:
        a = Act(length, chans, len(self.acts), persistent=persistent)
        self.acts.append(a)
        return a

    def vec(self, n: int) -> int:
        off = self.vec_len
        self.vec_len += (n + 3) // 4 * 4
        return off

    # ---------------- ops ---------------- #
    def _emit(self, words: Dict[int, int], reads, writes):
        op = [0] * OP_WORDS
        for k, v in words.items():
            op[k] = int(v)
        self.ops.append(op)
        self.op_acts.append((list(reads), writes))
        if len(self.op_items) < len(self.ops):
            self.op_items.append(None)

    def load_temb(self, n: int, dst_vec: int):
        self._emit({W_KIND: OP_LOAD_TEMB, L_NIN: n, L_DST: dst_vec}, [], None)

    def load_cond(self, n: int, dst_vec: int):
        """vec[dst : dst+n] <- this trajectory's raw condition features (zeros when the launch has none)."""
        self._emit({W_KIND: OP_LOAD_COND, L_NIN: n, L_DST: dst_vec}, [], None)

    def linear(self, lin_w: torch.Tensor, lin_b: torch.Tensor, src_vec: int, dst_vec: int, post_mish=False,
               raw_dst: Optional[int] = None):
        """dst = [Mish](W src + b); with `raw_dst` the pre-activation value is stored there as well."""
        n_out, n_in = lin_w.shape
        flags = (F_POST_MISH if post_mish else 0) | (F_RAW_COPY if raw_dst is not None else 0)
        self._emit({W_KIND: OP_LINEAR, L_NIN: n_in, L_NOUT: n_out, L_SRC: src_vec, L_DST: dst_vec,
                    L_WOFF: self.add(lin_w.t().contiguous()), L_BOFF: self.add(lin_b),
                    L_FLAGS: flags, L_DST2: raw_dst if raw_dst is not None else 0}, [], None)
        self.macs += n_in * n_out
        kparts = min(16, (N_WAVES * 64) // n_out)
        if kparts > 1:
            self.scratch = max(self.scratch, n_out * kparts)

    def fill(self, src_vec: int, n: int, dst: Act, coff: int):
        """Broadcast vec[src_vec : src_vec+n] into channels [coff, coff+n) of every row of a (persistent) slot."""
        self._emit({W_KIND: OP_FILL, L_NIN: n, L_NOUT: dst.length, L_SRC: src_vec, L_DST: 0, L_WOFF: dst.stride,
                    L_BOFF: coff}, [dst], None)

    def flatten(self, src: Act, dst_vec: int):
        self._emit({W_KIND: OP_FLATTEN, L_NIN: src.chans, L_NOUT: src.length, L_SRC: 0, L_DST: dst_vec,
                    L_WOFF: src.stride}, [src], None)

    def conv(self, srcs: Sequence[Act], dst: Act, w_eff: torch.Tensor, bias: torch.Tensor, *, stride=1, pad=0,
             transposed=False, gn: Optional[nn.Module] = None, emb_vec: int = -1, res: Optional[Act] = None,
             accum=False, dst_pred=False, act: Optional[int] = None, col_norm: Optional[nn.Module] = None,
             scale: Optional[float] = None, dst_coff: int = 0, keep_dst: bool = False, film: bool = False):
        """One fused conv/linear op.  Epilogue order: bias -> norm (slot-group `gn` | per-column `col_norm`) ->
        activation -> +FiLM vector -> +residual -> *scale -> store at channel offset `dst_coff`."""
        c_out, taps, _ = w_eff.shape
        assert not transposed or stride == 2, "the kernel's transposed-conv row map assumes stride 2"
        mode = MODE_4X4 if (dst.length <= 8 and c_out % 64 == 0 and self.allow_4x4) else MODE_16X16
        woff, n_chunks, per_src = self.pack_conv(w_eff, [s.chans for s in srcs], mode)
        rows = 16 if mode == MODE_16X16 else 64
        n_ct = -(-c_out // rows)
        ksplit = max(1, min(n_chunks, -(-N_WAVES // n_ct)))
        sstride = pad16(c_out) + 4
        items, qa = [], taps * per_src[0]
        for item in range(n_ct * ksplit):
            ct, ks = item % n_ct, item // n_ct
            q0, q1 = ks * n_chunks // ksplit, (ks + 1) * n_chunks // ksplit
            if q0 < qa:
                onb, tap, cc = 0, q0 // per_src[0], q0 % per_src[0]
            else:
                onb, tap, cc = 1, (q0 - qa) // per_src[1], (q0 - qa) % per_src[1]
            items.append([woff + (ct * n_chunks + q0) * 256, ks * dst.length * sstride + ct * rows, q1 - q0,
                          onb, tap, cc, 0, 0])
        self.op_items.append(items)
        flags = 0
        words = {W_KIND: OP_CONV, W_COUT: c_out, W_COUT16: pad16(c_out), W_LOUT: dst.length, W_TAPS: taps,
                 W_CSTRIDE: stride, W_CPAD: pad, W_TRANSPOSED: int(transposed),
                 W_SRCA: 0, W_SRCA_STRIDE: srcs[0].stride, W_CA_CHUNKS: per_src[0],
                 W_SRCB: 0, W_SRCB_STRIDE: 0, W_CB_CHUNKS: 0,
                 W_DST: 0, W_DST_STRIDE: dst.stride, W_DST_ROWS: dst.length + 2 * HALO,
                 W_WOFF: woff, W_BOFF: self.add(bias), W_KSPLIT: ksplit, W_NCHUNKS: n_chunks,
                 W_LIN: srcs[0].length, W_MODE: mode, W_NITEMS: len(items),
                 W_INV_COUT: _fbits(1.0 / c_out)}
        if len(srcs) == 2:
            assert srcs[1].length == srcs[0].length
            words[W_SRCB_STRIDE], words[W_CB_CHUNKS] = srcs[1].stride, per_src[1]
        if gn is not None:
            flags |= F_GN_MISH
            assert abs(gn.eps - GN_EPS) < 1e-12 and c_out % gn.num_groups == 0
            words[W_GROUPS] = gn.num_groups
            cg = c_out // gn.num_groups
            words[W_CG], words[W_CG_SHIFT] = cg, (cg.bit_length() - 1 if cg & (cg - 1) == 0 else -1)
            words[W_INV_CNT] = _fbits(1.0 / (cg * dst.length))
            words[W_GAMMA], words[W_BETA] = self.add(gn.weight), self.add(gn.bias)
        words[W_ACT] = (ACT_MISH if gn is not None else ACT_NONE) if act is None else act
        words[W_NORM] = NORM_SLOT_GROUP if gn is not None else NORM_NONE
        if col_norm is not None:
            assert gn is None and abs(col_norm.eps - GN_EPS) < 1e-12
            groups = getattr(col_norm, "num_groups", 1)             # nn.LayerNorm == one group over all channels
            assert c_out % groups == 0
            cg = c_out // groups
            assert c_out <= 1024, "per-column norm: C_out <= 1024"       # any group size: the epilogue strides lanes over cg
            words[W_NORM], words[W_GROUPS], words[W_CG] = NORM_COLUMN, groups, cg
            words[W_CG_SHIFT] = cg.bit_length() - 1 if cg & (cg - 1) == 0 else -1
            words[W_INV_CNT] = _fbits(1.0 / cg)
            words[W_GAMMA], words[W_BETA] = self.add(col_norm.weight), self.add(col_norm.bias)
        if scale is not None:
            flags |= F_SCALE
            words[W_SCALE] = _fbits(scale)
        if keep_dst or dst_coff:
            flags |= F_KEEP_DST
        words[W_DST_COFF] = dst_coff
        if emb_vec >= 0:
            flags |= F_FILM if film else F_ADD_EMB      # FiLM: y*e[c] + e[C + c]; else y + e[c]
            words[W_EMB] = emb_vec
        if res is not None:
            flags |= F_ADD_RES
            assert res.chans >= c_out and res.length == dst.length
            words[W_RES_STRIDE] = res.stride
        if accum:
            flags |= F_ACCUM
        if dst_pred:
            flags |= F_DST_PRED
        words[W_FLAGS] = flags
        reads = list(srcs) + ([res] if res is not None else []) + ([dst] if accum else [])
        self._emit(words, reads, dst)
        self.scratch = max(self.scratch, ksplit * dst.length * (pad16(c_out) + 4))
        self.macs += c_out * dst.length * taps * sum(s.chans for s in srcs) // (2 if transposed else 1)
        self.n_conv += 1

    # ---------------- LDS planning ---------------- #
    def plan_lds(self, fixed: Dict[str, int]) -> int:
        """Linear-scan interval allocation of the activation arena; patches slot offsets into the ops."""
        n = len(self.ops)
        first, last = {}, {}
        for i, (reads, writes) in enumerate(self.op_acts):
            for a in reads + ([writes] if writes is not None else []):
                first.setdefault(a.uid, i)
                last[a.uid] = i
        base = fixed["arena"]
        live: List[Act] = []
        top = base
        for a in self.acts:
            if a.persistent:
                continue
            if a.uid not in first:
                continue
        order = sorted((a for a in self.acts if not a.persistent and a.uid in first), key=lambda a: first[a.uid])
        for a in order:
            t = first[a.uid]
            live = [b for b in live if last[b.uid] >= t]
            # first-fit among gaps between live slots
            spans = sorted((b.off, b.off + b.floats) for b in live)
            pos = base
            for lo, hi in spans:
                if lo - pos >= a.floats:
                    break
                pos = max(pos, hi)
            a.off = pos
            live.append(a)
            top = max(top, pos + a.floats)
        # patch offsets
        for op, (reads, writes) in zip(self.ops, self.op_acts):
            if op[W_KIND] == OP_FLATTEN:
                op[L_SRC] = reads[0].off
            if op[W_KIND] == OP_FILL:
                op[L_DST] = reads[0].off
            if op[W_KIND] != OP_CONV:
                continue
            srcs = reads[:2] if op[W_CB_CHUNKS] else reads[:1]
            op[W_SRCA] = srcs[0].off
            if op[W_CB_CHUNKS]:
                op[W_SRCB] = srcs[1].off
            op[W_DST] = writes.off
            if op[W_FLAGS] & F_ADD_RES:
                op[W_RES] = reads[len(srcs)].off
        return top


# ================================================================================================== #
# JannerUNet1d lowering                                                                              #
# ================================================================================================== #

def _conv1d_eff(conv: nn.Conv1d) -> torch.Tensor:
    return conv.weight.detach().permute(0, 2, 1)            # (C_out, C_in, k) -> [co][tap][ci]


def _convT1d_eff(conv: nn.ConvTranspose1d) -> torch.Tensor:
    return conv.weight.detach().permute(1, 2, 0)            # (C_in, C_out, k) -> [co][tap][ci]


def supports_janner(net) -> Optional[str]:
    """None if the fused kernel can run this JannerUNet1d, else the reason it cannot."""
    if net.attention:
        return "attention=True (LinearAttention) is PyTorch-only"
    if net.norm_type != "groupnorm":
        return f"norm_type={net.norm_type!r} is PyTorch-only"
    if net.kernel_size % 2 == 0:
        return f"kernel_size={net.kernel_size} unsupported (odd only)"
    return None


def _resblock(b: "_Builder", srcs: List[Act], rb, k: int, emb_vec: int) -> Act:
    """ResidualBlock (reference jannerunet.py:51-69) = 2 fused conv ops (+1 accumulate op for a 1x1 skip conv)."""
    c_out = rb.conv1[0].out_channels
    length = srcs[0].length
    t1 = b.act(length, c_out)
    b.conv(srcs, t1, _conv1d_eff(rb.conv1[0]), rb.conv1[0].bias, pad=k // 2, gn=rb.conv1[1], emb_vec=emb_vec)
    out = b.act(length, c_out)
    identity = isinstance(rb.residual_conv, nn.Identity)
    if identity:
        assert len(srcs) == 1
    b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1],
           res=srcs[0] if identity else None)
    if not identity:
        b.conv(srcs, out, _conv1d_eff(rb.residual_conv), rb.residual_conv.bias, accum=True)
    return out


def _downsample(b: "_Builder", cur: Act, down) -> Act:
    nxt = b.act((cur.length - 1) // 2 + 1, cur.chans)          # Conv1d(k=3, stride=2, pad=1)
    b.conv([cur], nxt, _conv1d_eff(down.conv), down.conv.bias, stride=2, pad=1)
    return nxt


def _emb_chain(b: "_Builder", net, blocks, raw_emb_vec: Optional[int] = None):
    """temb(+cond) -> Linear -> Mish -> Linear -> Mish -> stacked per-block FiLM Linear.  Returns block -> vec offset.
    `emb` is consumed by the blocks only through ``emb_mlp = Mish -> Linear``, so Mish(emb) is what gets stored;
    callers that also need the raw embedding (classifier head) pass `raw_emb_vec`."""
    md = net.model_dim
    v_temb, v_hid, v_memb = b.vec(net.emb_dim), b.vec(md * 4), b.vec(md)
    slices, total = {}, 0
    for rb in blocks:
        slices[id(rb)] = total
        total += rb.emb_mlp[1].out_features
    v_eall = b.vec(total)
    b.load_temb(net.emb_dim, v_temb)
    b.linear(net.map_emb[0].weight, net.map_emb[0].bias, v_temb, v_hid, post_mish=True)
    b.linear(net.map_emb[2].weight, net.map_emb[2].bias, v_hid, v_memb, post_mish=True, raw_dst=raw_emb_vec)
    b.linear(torch.cat([rb.emb_mlp[1].weight for rb in blocks], 0),
             torch.cat([rb.emb_mlp[1].bias for rb in blocks], 0), v_memb, v_eall)
    return {k: v_eall + v for k, v in slices.items()}


def _finalize(b: "_Builder", net, x: Act, pred: Optional[Act], horizon: int, d: int, max_lds_bytes: int,
              out_vec: int = -1, out_len: int = 0, persist: Sequence[Act] = (), emb_dim: Optional[int] = None,
              cond_slot: Optional[Tuple[Act, int, int]] = None, tile: int = 0,
              vec_alias: Sequence[Tuple[Act, int]] = (), edm: bool = True) -> Program:
    """LDS map [x | pred0 | pred1 | prev(dense) | vec | scratch | descriptors | stamps | arena...], offsets patched.
    `edm`: reserve the two extra dense buffers (x_old, x_true) the EDM / consistency step kinds keep next to the multistep
    memory; programs compiled without them are 2 x H x D floats smaller and must only be launched with step kinds 0-4."""
    dev = b.device
    off = 0
    x.off, off = off, off + x.floats
    pred_off = pred_stride = pred_branch = 0
    if pred is not None:
        pred.off, off = off, off + pred.floats
        pred_off, pred_stride, pred_branch = pred.off, pred.stride, pred.floats
        off += pred.floats                                # second prediction slot (CFG unconditional branch)
    for a in persist:                                     # further kernel-lifetime slots (MLP context etc.)
        a.off, off = off, off + a.floats
    # shared zero row: what a conv tap outside [0, L) reads (and columns past l_out); sized for the widest source
    zrow_off = off
    sources = [a for reads, _ in b.op_acts for a in reads]
    zrow_floats = max(pad16(a.chans) for a in sources) + 16
    off += zrow_floats
    prev_off, off = off, off + (3 if edm else 1) * ((horizon * d + 3) // 4 * 4)     # multistep memory / EDM slope [| x_old | x_true]
    vec_off, off = off, off + b.vec_len
    for a, rel in vec_alias:                              # 1-row slots that ARE vectors (Linear lowered as a 1-position conv)
        a.off = vec_off + rel
    scratch_off, off = off, off + (b.scratch + 3) // 4 * 4
    op_words = len(b.ops) * OP_WORDS
    all_words = op_words + sum(len(it) * ITEM_WORDS for it in b.op_items if it)
    # the item tables ride in LDS with the ops unless that would blow the budget (big programs keep them in HBM)
    items_in_lds = all_words <= 6144
    desc_words = all_words if items_in_lds else op_words
    desc_off, off = off, off + (desc_words + 3) // 4 * 4
    prof_off, off = off, off + (2 * (len(b.ops) * 8 + 2) + 3) // 4 * 4
    top = b.plan_lds({"arena": off})
    if top * 4 > max_lds_bytes:
        raise ValueError(f"LDS plan needs {top * 4} B > {max_lds_bytes} B (horizon {horizon} too long for one workgroup)")
    # item tables live behind the ops in the same buffer; W_ITEMS = word offset from the buffer start
    tail, cursor = [], len(b.ops) * OP_WORDS
    for op, items in zip(b.ops, b.op_items):
        if items:
            op[W_ITEMS] = cursor
            tail += [w for rec in items for w in rec]
            cursor += len(items) * ITEM_WORDS
    ops = np.asarray(b.ops, dtype=np.int32)
    for op in ops:                                        # vec offsets were relative; make them absolute
        if op[W_KIND] in (OP_LOAD_TEMB, OP_LOAD_COND):
            op[L_DST] += vec_off
        elif op[W_KIND] == OP_LINEAR:
            op[L_SRC] += vec_off
            op[L_DST] += vec_off
            if op[L_FLAGS] & F_RAW_COPY:
                op[L_DST2] += vec_off
        elif op[W_KIND] == OP_FLATTEN:
            op[L_DST] += vec_off
        elif op[W_KIND] == OP_FILL:
            op[L_SRC] += vec_off
        elif op[W_KIND] == OP_CONV and op[W_FLAGS] & (F_ADD_EMB | F_FILM):
            op[W_EMB] += vec_off
    blob = torch.cat(b.chunks) if b.chunks else torch.zeros(0, device=dev)
    ops_buffer = np.concatenate([ops.reshape(-1), np.asarray(tail, dtype=np.int64).astype(np.int32)])
    assert ops_buffer.size == all_words
    return Program(ops=ops, ops_buffer=ops_buffer, blob=blob.contiguous(), lds_floats=top, x_off=x.off,
                   x_stride=x.stride, pred_off=pred_off, pred_stride=pred_stride, pred_branch_floats=pred_branch,
                   prev_off=prev_off, vec_off=vec_off, scratch_off=scratch_off, scratch_floats=b.scratch,
                   desc_off=desc_off, desc_words=desc_words, items_in_lds=items_in_lds, prof_off=prof_off,
                   horizon=horizon, dim=d,
                   emb_dim=net.emb_dim if emb_dim is None else emb_dim, tile=tile,
                   cond_slot_off=cond_slot[0].off if cond_slot else 0,
                   cond_slot_stride=cond_slot[0].stride if cond_slot else 0,
                   cond_coff=cond_slot[1] if cond_slot else 0, cond_dim=cond_slot[2] if cond_slot else 0,
                   persist_slots=[(a.off, a.floats) for a in persist],
                   zero_off=persist[0].off if persist else zrow_off,
                   zero_floats=sum(a.floats for a in persist) + zrow_floats, zrow_off=zrow_off,
                   macs_per_forward=b.macs, n_conv=b.n_conv,
                   out_vec_off=(vec_off + out_vec) if out_vec >= 0 else 0, out_vec_len=out_len,
                   meta={"n_ops": len(ops), "blob_floats": int(blob.numel())})


def compile_janner(net, horizon: int, max_lds_bytes: int = 160 * 1024, allow_4x4: bool = True, edm: bool = True) -> Program:
    """Lower a JannerUNet1d (reference nn_diffusion/jannerunet.py:98-201 structure) for `horizon` positions."""
    why = supports_janner(net)
    if why is not None:
        raise ValueError(why)
    b = _Builder(next(net.parameters()).device)
    b.allow_4x4 = allow_4x4
    d, k, md = net.in_dim, net.kernel_size, net.model_dim

    blocks = []
    for res1, res2, _, _ in net.downs:
        blocks += [res1, res2]
    blocks += [net.mid_block1, net.mid_block2]
    for res1, res2, _, _ in net.ups:
        blocks += [res1, res2]
    emb_of = _emb_chain(b, net, blocks)

    def resblock(srcs, rb):
        return _resblock(b, srcs, rb, k, emb_of[id(rb)])

    x = b.act(horizon, d, persistent=True)
    cur, skips = x, []
    for res1, res2, _, down in net.downs:
        cur = resblock([resblock([cur], res1)], res2)
        skips.append(cur)
        if not isinstance(down, nn.Identity):
            assert cur.length % 2 == 0, "horizon too short for the number of resolutions"
            cur = _downsample(b, cur, down)
    cur = resblock([resblock([cur], net.mid_block1)], net.mid_block2)
    for res1, res2, _, up in net.ups:
        cur = resblock([resblock([cur, skips.pop()], res1)], res2)
        if not isinstance(up, nn.Identity):
            nxt = b.act(cur.length * 2, cur.chans)
            b.conv([cur], nxt, _convT1d_eff(up.conv), up.conv.bias, stride=2, pad=1, transposed=True)
            cur = nxt
    assert cur.length == horizon
    fc = net.final_conv
    t = b.act(horizon, md)
    b.conv([cur], t, _conv1d_eff(fc[0]), fc[0].bias, pad=2, gn=fc[1])
    pred = b.act(horizon, d, persistent=True)
    b.conv([t], pred, _conv1d_eff(fc[3]), fc[3].bias, dst_pred=True)
    return _finalize(b, net, x, pred, horizon, d, max_lds_bytes, edm=edm)


def supports_half_janner(net) -> Optional[str]:
    if net.norm_type != "groupnorm":
        return f"norm_type={net.norm_type!r} is PyTorch-only"
    if net.kernel_size % 2 == 0:
        return f"kernel_size={net.kernel_size} unsupported (odd only)"
    return None


def compile_half_janner(net, horizon: int, max_lds_bytes: int = 160 * 1024, allow_4x4: bool = True) -> Program:
    """Lower a HalfJannerUNet1d classifier (reference nn_classifier/half_jannerunet.py:11-125): encoder half, two
    k=5 mid blocks each followed by a stride-2 conv, channel-major flatten, concat with the raw time embedding,
    Linear-Mish-Linear head.  Output = a (out_dim,) vector per trajectory (``Program.out_vec_*``)."""
    why = supports_half_janner(net)
    if why is not None:
        raise ValueError(why)
    if horizon != net.horizon:
        raise ValueError(f"HalfJannerUNet1d was built for horizon {net.horizon}, got {horizon}")
    b = _Builder(next(net.parameters()).device)
    b.allow_4x4 = allow_4x4
    d, k, md = net.in_dim, net.kernel_size, net.model_dim

    blocks = []
    for res1, res2, _ in net.downs:
        blocks += [res1, res2]
    blocks += [net.mid_block1[0], net.mid_block2[0]]
    head_in = net.final_block[0].in_features
    fc_dim = head_in - md
    v_cat = b.vec(head_in)                                   # [flatten(x) | raw emb]
    emb_of = _emb_chain(b, net, blocks, raw_emb_vec=v_cat + fc_dim)

    x = b.act(horizon, d, persistent=True)
    cur = x
    for res1, res2, down in net.downs:
        cur = _resblock(b, [_resblock(b, [cur], res1, k, emb_of[id(res1)])], res2, k, emb_of[id(res2)])
        if not isinstance(down, nn.Identity):
            cur = _downsample(b, cur, down)
    for blk, down in (net.mid_block1, net.mid_block2):
        cur = _downsample(b, _resblock(b, [cur], blk, 5, emb_of[id(blk)]), down)
    assert cur.chans * cur.length == fc_dim, (cur.chans, cur.length, fc_dim)
    b.flatten(cur, v_cat)
    v_h, v_out = b.vec(net.final_block[0].out_features), b.vec(net.out_dim)
    b.linear(net.final_block[0].weight, net.final_block[0].bias, v_cat, v_h, post_mish=True)
    b.linear(net.final_block[2].weight, net.final_block[2].bias, v_h, v_out)
    return _finalize(b, net, x, None, horizon, d, max_lds_bytes, out_vec=v_out, out_len=net.out_dim, edm=False)   # forward-only


# ================================================================================================== #
# Batch-tiled MLP denoisers: one workgroup = `tile` samples, the sample index rides the MFMA column   #
# axis (a Linear is a 1-tap conv over `tile` "positions"), the whole sampling loop stays in the launch #
# ================================================================================================== #
MLP_TILE = 16


def _lin_eff(lin: nn.Linear, pad_in: int = 0) -> torch.Tensor:
    """(n_out, n_in) -> [co][tap=1][ci], optionally zero-padding extra trailing input channels."""
    w = lin.weight.detach()
    if pad_in:
        w = torch.cat([w, torch.zeros(w.shape[0], pad_in, device=w.device, dtype=w.dtype)], 1)
    return w.unsqueeze(1)


def compile_pearce_mlp(net, tile: int = MLP_TILE, max_lds_bytes: int = 160 * 1024, edm: bool = True) -> Program:
    """PearceMlp (reference nn_diffusion/pearcemlp.py:36-79).  Slots: state [x | raw t] (D+1 channels), context
    [t_emb | flattened condition]; FCBlock = Linear -> per-sample GroupNorm -> GELU(erf); skips are stored pre-scaled by
    1/1.414 exactly where the reference divides (Q11)."""
    b = _Builder(next(net.parameters()).device)
    d, e, hd, n_cond = net.act_dim, net.emb_dim, net.hidden_dim, net.To * net.emb_dim
    s = 1.0 / net.SKIP_SCALE
    x = b.act(tile, d + 1, persistent=True)
    ctx = b.act(tile, e + n_cond, persistent=True)
    v_temb = b.vec(e + 1)                                   # [map_noise(t) | float(t)] row of the host table
    b.load_temb(e + 1, v_temb)
    b.fill(v_temb, e, ctx, 0)
    b.fill(v_temb + e, 1, x, d)
    a1, xe = b.act(tile, e), b.act(tile, e)
    b.conv([x], a1, _lin_eff(net.act_emb[0], pad_in=1), net.act_emb[0].bias, act=ACT_LEAKY)
    b.conv([a1], xe, _lin_eff(net.act_emb[2]), net.act_emb[2].bias)
    h1, h2, h3 = b.act(tile, hd), b.act(tile, hd), b.act(tile, hd)
    f = net.fcs
    b.conv([xe, ctx], h1, _lin_eff(f[0].model[0]), f[0].model[0].bias, col_norm=f[0].model[1], act=ACT_GELU_ERF, scale=s)
    b.conv([h1, x], h2, _lin_eff(f[1].model[0]), f[1].model[0].bias, col_norm=f[1].model[1], act=ACT_GELU_ERF,
           res=h1, scale=s)
    b.conv([h2, x], h3, _lin_eff(f[2].model[0]), f[2].model[0].bias, col_norm=f[2].model[1], act=ACT_GELU_ERF, res=h2)
    pred = b.act(tile, d, persistent=True)
    b.conv([h3, x], pred, _lin_eff(f[3]), f[3].bias, dst_pred=True)
    return _finalize(b, net, x, pred, tile, d, max_lds_bytes, persist=[ctx], emb_dim=e + 1,
                     cond_slot=(ctx, e, n_cond), tile=tile, edm=edm)


def compile_dql_mlp(net, tile: int = MLP_TILE, max_lds_bytes: int = 160 * 1024, edm: bool = True) -> Program:
    """DQLMlp (reference nn_diffusion/dqlmlp.py:9-52) and DVInvMlp (dvinvmlp.py:9-47, same trunk): features [x | time_mlp(map_noise(t)) | obs] -> 3 x (Linear, Mish)
    -> Linear.  The time MLP is batch-invariant, so it runs once per step on a vector and is broadcast into the context."""
    b = _Builder(next(net.parameters()).device)
    d = net.final_layer.out_features
    e = net.time_mlp[0].in_features
    obs = net.obs_dim
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, e + obs, persistent=True)
    v0, v1, v2 = b.vec(e), b.vec(2 * e), b.vec(e)
    b.load_temb(e, v0)
    b.linear(net.time_mlp[0].weight, net.time_mlp[0].bias, v0, v1, post_mish=True)
    b.linear(net.time_mlp[2].weight, net.time_mlp[2].bias, v1, v2)
    b.fill(v2, e, ctx, 0)
    m = net.mid_layer
    hid = m[0].out_features                               # 256 for DQLMlp, configurable for DVInvMlp
    m1, m2, m3 = b.act(tile, hid), b.act(tile, hid), b.act(tile, hid)
    b.conv([x, ctx], m1, _lin_eff(m[0]), m[0].bias, act=ACT_MISH)
    b.conv([m1], m2, _lin_eff(m[2]), m[2].bias, act=ACT_MISH)
    b.conv([m2], m3, _lin_eff(m[4]), m[4].bias, act=ACT_MISH)
    pred = b.act(tile, d, persistent=True)
    b.conv([m3], pred, _lin_eff(net.final_layer), net.final_layer.bias, dst_pred=True)
    return _finalize(b, net, x, pred, tile, d, max_lds_bytes, persist=[ctx], emb_dim=e,
                     cond_slot=(ctx, e, obs), tile=tile, edm=edm)


def _act_id(m) -> Optional[int]:
    """nn activation module -> ACT_* id of the program kernel, None when it has no native epilogue."""
    if isinstance(m, nn.ReLU):
        return ACT_RELU
    if isinstance(m, nn.Mish):
        return ACT_MISH
    if isinstance(m, nn.SiLU):
        return ACT_SILU
    if isinstance(m, nn.Tanh):
        return ACT_TANH
    if isinstance(m, nn.GELU):
        return ACT_GELU_TANH if getattr(m, "approximate", "none") == "tanh" else ACT_GELU_ERF
    if isinstance(m, nn.LeakyReLU) and abs(m.negative_slope - 0.01) < 1e-12:
        return ACT_LEAKY
    if isinstance(m, nn.Identity):
        return ACT_NONE
    return None


def compile_mlp_nn(net, tile: int = MLP_TILE, max_lds_bytes: int = 160 * 1024, edm: bool = True) -> Program:
    """MlpNNDiffusion (reference nn_diffusion/mlps.py:10-40): Mlp(cat[x, map_noise(t) + condition]) -- hidden Linears with one
    activation, an output Linear.  The context slot holds [map_noise(t) | condition]; the first layer's time columns are applied
    to both halves."""
    b = _Builder(next(net.parameters()).device)
    layers = list(net.mlp.mlp)
    lins = [m[0] if isinstance(m, nn.Sequential) else m for m in layers if isinstance(m, (nn.Sequential, nn.Linear))]
    acts = [_act_id(m[1]) for m in layers if isinstance(m, nn.Sequential)] + [_act_id(layers[-1])]
    if any(a is None for a in acts) or len(lins) != len(acts):
        raise ValueError("MlpNNDiffusion: activation without a native epilogue")
    d = lins[-1].out_features
    e = lins[0].in_features - d
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, 2 * e, persistent=True)
    v0 = b.vec(e)
    b.load_temb(e, v0)
    b.fill(v0, e, ctx, 0)
    w0 = lins[0].weight.detach()
    w_first = torch.cat([w0, w0[:, d:]], 1).unsqueeze(1)          # [W_x | W_t | W_t]: the condition is ADDED to the time embedding
    cur, srcs = None, [x, ctx]
    for i, (lin, act) in enumerate(zip(lins[:-1], acts[:-1])):
        nxt = b.act(tile, lin.out_features)
        b.conv(srcs, nxt, w_first if i == 0 else _lin_eff(lin), lin.bias, act=act)
        cur, srcs = nxt, [nxt]
    pred = b.act(tile, d, persistent=True)
    b.conv(srcs, pred, w_first if len(lins) == 1 else _lin_eff(lins[-1]), lins[-1].bias, dst_pred=True, act=acts[-1])
    return _finalize(b, net, x, pred, tile, d, max_lds_bytes, persist=[ctx], emb_dim=e, cond_slot=(ctx, e, e), tile=tile, edm=edm)


def compile_sfbc_unet(net, tile: int = MLP_TILE, max_lds_bytes: int = 160 * 1024, edm: bool = True) -> Program:
    """SfBCUNet (reference nn_diffusion/sfbc_unet.py:9-82): residual blocks of Linears, block(x, c) = SiLU(L2(SiLU(L1 x) + Lc c)) +
    skip(x), a down path, a middle block, an up path over the concat with the matching down activation, one output Linear.
    The context c = t_layer(map_noise(t)) + condition is never formed: the context slot holds [t_layer(...) | condition] side by
    side and Lc is applied to both halves ([Lc | Lc]); the t_layer table (batch-invariant) is the launch's `temb`."""
    b = _Builder(next(net.parameters()).device)
    d, e = net.out_layer.out_features, net.t_layer[0].in_features
    x = b.act(tile, d, persistent=True)
    ctx = b.act(tile, 2 * e, persistent=True)
    v0 = b.vec(e)
    b.load_temb(e, v0)
    b.fill(v0, e, ctx, 0)

    def block(srcs, blk):
        c_out = blk.linear1[0].out_features
        h = b.act(tile, c_out)
        b.conv(srcs, h, _lin_eff(blk.linear1[0]), blk.linear1[0].bias, act=ACT_SILU)
        wc = blk.linearc.weight.detach()
        b.conv([ctx], h, torch.cat([wc, wc], 1).unsqueeze(1), blk.linearc.bias, accum=True, act=ACT_NONE)      # h += Lc (t + cond)
        o = b.act(tile, c_out)
        ident = isinstance(blk.skip, nn.Identity)
        b.conv([h], o, _lin_eff(blk.linear2[0]), blk.linear2[0].bias, act=ACT_SILU, res=srcs[0] if ident and len(srcs) == 1 else None)
        if not ident:
            b.conv(srcs, o, _lin_eff(blk.skip), blk.skip.bias, accum=True, act=ACT_NONE)
        elif len(srcs) > 1:               # identity skip over a concat (up block whose widths happen to match): o += cat(srcs)
            eye = torch.eye(c_out, device=b.device).unsqueeze(1)
            b.conv(srcs, o, eye, torch.zeros(c_out, device=b.device), accum=True, act=ACT_NONE)
        return o

    cur, kept = x, []
    for blk in net.down_blocks:
        cur = block([cur], blk)
        kept.append(cur)
    cur = block([cur], net.mid_block)
    for blk in net.up_blocks:
        cur = block([cur, kept.pop()], blk)
    pred = b.act(tile, d, persistent=True)
    b.conv([cur], pred, _lin_eff(net.out_layer), net.out_layer.bias, dst_pred=True, act=ACT_NONE)
    return _finalize(b, net, x, pred, tile, d, max_lds_bytes, persist=[ctx], emb_dim=e, cond_slot=(ctx, e, e), tile=tile, edm=edm)


# ================================================================================================== #
# ChiUNet1d lowering (Diffusion Policy)                                                               #
# ================================================================================================== #
def supports_chiunet(net) -> Optional[str]:
    if not net.obs_as_global_cond:
        return "local (per-timestep) observation conditioning is PyTorch-only"
    k = net.final_conv[0].kernel_size[0]
    if k % 2 == 0:
        return f"kernel_size={k} unsupported (odd only)"
    return None


def compile_chiunet(net, horizon: int, max_lds_bytes: int = 160 * 1024, allow_4x4: bool = True, edm: bool = True) -> Program:
    """ChiUNet1d with a global condition (reference nn_diffusion/chiunet.py:48-192).  The FiLM vector of a block
    (Mish -> Linear(2*emb -> [2]C)) is computed just in time into one reusable vec region -- at config-3 width the
    stacked vectors of all blocks would not fit LDS."""
    why = supports_chiunet(net)
    if why is not None:
        raise ValueError(why)
    b = _Builder(next(net.parameters()).device)
    b.allow_4x4 = allow_4x4
    d = net.final_conv[3].out_channels
    k = net.final_conv[0].kernel_size[0]
    e = net.emb_dim
    n_cond = net.global_cond_encoder.in_features
    blocks = []
    for res1, res2, _ in net.downs:
        blocks += [res1, res2]
    blocks += list(net.mids)
    for res1, res2, _ in net.ups:
        blocks += [res1, res2]
    # emb = [map_emb(temb) | global_cond_encoder(cond)]; blocks consume Mish(emb) only
    v_temb, v_hid, v_cond, v_memb = b.vec(e), b.vec(4 * e), b.vec(n_cond), b.vec(2 * e)
    v_film = b.vec(max(blk.cond_encoder[1].out_features for blk in blocks))
    b.load_temb(e, v_temb)
    b.load_cond(n_cond, v_cond)
    b.linear(net.map_emb[0].weight, net.map_emb[0].bias, v_temb, v_hid, post_mish=True)
    b.linear(net.map_emb[2].weight, net.map_emb[2].bias, v_hid, v_memb, post_mish=True)
    b.linear(net.global_cond_encoder.weight, net.global_cond_encoder.bias, v_cond, v_memb + e, post_mish=True)

    # The per-block FiLM Linear (2*emb -> [2]C, up to 4 MB of weights) is streamed like a conv: the Mish(emb) vector is
    # viewed as a 1-position slot, so the weights arrive as 1-KiB MFMA records through the prefetch ring instead of
    # 4-byte-per-lane loads (measured: the scalar form cost ~30 % of a config-3 forward).
    memb_slot = b.act(1, 2 * e, persistent=True)
    film_slot = b.act(1, max(blk.cond_encoder[1].out_features for blk in blocks), persistent=True)

    def resblock(srcs: List[Act], rb) -> Act:
        c_out, length = rb.out_dim, srcs[0].length
        enc = rb.cond_encoder[1]
        b.conv([memb_slot], film_slot, _lin_eff(enc), enc.bias, keep_dst=True)
        t1 = b.act(length, c_out)
        b.conv(srcs, t1, _conv1d_eff(rb.conv1[0]), rb.conv1[0].bias, pad=k // 2, gn=rb.conv1[1], emb_vec=v_film,
               film=rb.cond_predict_scale)
        out = b.act(length, c_out)
        identity = isinstance(rb.residual_conv, nn.Identity)
        if identity:
            assert len(srcs) == 1
        b.conv([t1], out, _conv1d_eff(rb.conv2[0]), rb.conv2[0].bias, pad=k // 2, gn=rb.conv2[1],
               res=srcs[0] if identity else None)
        if not identity:
            b.conv(srcs, out, _conv1d_eff(rb.residual_conv), rb.residual_conv.bias, accum=True)
        return out

    x = b.act(horizon, d, persistent=True)
    cur, skips = x, []
    for res1, res2, down in net.downs:
        cur = resblock([resblock([cur], res1)], res2)
        skips.append(cur)
        if not isinstance(down, nn.Identity):
            assert cur.length % 2 == 0, "horizon too short for the number of resolutions"
            cur = _downsample(b, cur, down)
    for mid in net.mids:
        cur = resblock([cur], mid)
    for res1, res2, up in net.ups:
        cur = resblock([resblock([cur, skips.pop()], res1)], res2)
        if not isinstance(up, nn.Identity):
            nxt = b.act(cur.length * 2, cur.chans)
            b.conv([cur], nxt, _convT1d_eff(up.conv), up.conv.bias, stride=2, pad=1, transposed=True)
            cur = nxt
    assert cur.length == horizon
    fc = net.final_conv
    t = b.act(horizon, net.model_dim)
    b.conv([cur], t, _conv1d_eff(fc[0]), fc[0].bias, pad=k // 2, gn=fc[1])
    pred = b.act(horizon, d, persistent=True)
    b.conv([t], pred, _conv1d_eff(fc[3]), fc[3].bias, dst_pred=True)
    prog = _finalize(b, net, x, pred, horizon, d, max_lds_bytes, emb_dim=e,
                     vec_alias=[(memb_slot, v_memb), (film_slot, v_film)], edm=edm)
    prog.cond_dim = n_cond
    return prog
