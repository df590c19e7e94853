"""Lane-level numpy model of the program kernel ``csrc/cdx_unet2.hip`` -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Interprets the SAME (ops, item tables, blob, LDS plan) the v2 kernel receives with the kernel's index arithmetic:
items per wave, the (segment, tap, chunk) cursor, the 64-lane operand
fetch of the two MFMA shapes, the staging layout, and the epilogue's thread -> (group, float4 item) partition.  It is the
executable specification ``cdx_unet2.hip`` was transcribed from and the CPU proof that the compiler's packing / slot
plan / item tables are right before a GPU minute is spent.  One instance = one trajectory (trajectories never interact).
"""
import numpy as np

from cleandiffuser_amd.engine import consts as P
from cleandiffuser_amd.engine import program2 as P2
from cleandiffuser_amd.engine.consts import GN_EPS, MODE_16X16


def mish(x):
    x = np.asarray(x, np.float32)
    e = np.exp(np.minimum(x, 20.0).astype(np.float32))
    n = e * (e + 2.0)
    return np.where(x > 20.0, x, x * n / (n + 2.0)).astype(np.float32)


def activation(v, act_id):
    import math
    v = np.asarray(v, np.float32)
    if act_id == P.ACT_NONE:
        return v
    if act_id == P.ACT_MISH:
        return mish(v)
    if act_id == P.ACT_GELU_ERF:
        erf = np.vectorize(math.erf)
        return (0.5 * v * (1.0 + erf(v.astype(np.float64) / math.sqrt(2.0)))).astype(np.float32)
    if act_id == P.ACT_LEAKY:
        return np.where(v > 0, v, np.float32(0.01) * v).astype(np.float32)
    if act_id == P.ACT_SILU:
        return (v / (1.0 + np.exp(-v))).astype(np.float32)
    if act_id == P.ACT_RELU:
        return np.maximum(v, 0).astype(np.float32)
    if act_id == P.ACT_GELU_TANH:
        return (0.5 * v * (1.0 + np.tanh(0.7978845608028654 * (v + 0.044715 * v ** 3)))).astype(np.float32)
    if act_id == P.ACT_TANH:
        return np.tanh(v).astype(np.float32)
    raise ValueError(act_id)


def mish_grad(a):
    """d Mish(a) / d a in the kernel's form: n/(n+2) + a 4 e (e+1) / (n+2)^2, e = exp(min(a, 20)), n = e (e + 2)."""
    a = np.asarray(a, np.float32)
    e = np.exp(np.minimum(a, np.float32(20.0))).astype(np.float32)
    n = (e * (e + np.float32(2.0))).astype(np.float32)
    r = (np.float32(1.0) / (n + np.float32(2.0))).astype(np.float32)
    return (n * r + a * (np.float32(4.0) * e * (e + np.float32(1.0))) * (r * r)).astype(np.float32)


def emb_table(prog: P2.Program2, temb, tembs=None) -> np.ndarray:
    """What cdx_unet2_embtab computes per network of the program: rows of Linear(Mish(map_emb(temb_row))) for every block
    (+ the rows applied to the RAW embedding: the classifier head), (steps, n_emb).  `tembs`: one map_noise(t) array per network."""
    blob = prog.blob.detach().cpu().numpy()
    tembs = [temb] if tembs is None else list(tembs)
    specs = prog.embtabs if prog.embtabs else [prog.embtab]
    assert len(tembs) == len(specs)
    out = np.zeros((np.asarray(tembs[0]).shape[0], prog.n_emb), np.float32)
    for e, tb in zip(specs, tembs):
        ed, hid, md, n = e["emb_dim"], e["hidden"], e["md"], e["n_emb"]
        w0 = blob[e["w0"]:e["w0"] + ed * hid].reshape(ed, hid)
        w2 = blob[e["w2"]:e["w2"] + hid * md].reshape(hid, md)
        w3 = blob[e["w3"]:e["w3"] + md * n].reshape(md, n)
        for s, row in enumerate(np.asarray(tb, np.float32)):
            h = blob[e["b0"]:e["b0"] + hid].copy()
            for i in range(ed):
                h = h + w0[i] * row[i]
            h = mish(h)
            raw = blob[e["b2"]:e["b2"] + md].copy()
            for i in range(hid):
                raw = raw + w2[i] * h[i]
            m = mish(raw)
            o = blob[e["b3"]:e["b3"] + n].copy()
            for i in range(md):
                o = o + w3[i] * m[i]
            out[s, e.get("col0", 0):e.get("col0", 0) + n] = o
            if e.get("n_raw", 0):
                w4 = blob[e["w4"]:e["w4"] + md * e["n_raw"]].reshape(md, e["n_raw"])
                o4 = blob[e["b4"]:e["b4"] + e["n_raw"]].copy()
                for i in range(md):
                    o4 = o4 + w4[i] * raw[i]
                out[s, e["col4"]:e["col4"] + e["n_raw"]] = o4
    return out


class LaneSim2:
    def __init__(self, prog: P2.Program2, member: int = -1):
        """`member` >= 0: member view of a split program (one trajectory over k workgroups): this instance runs that member's
        descriptors; `run_forward_split` steps the k instances in lockstep and performs the exchanges."""
        self.p = prog
        self.member = member
        self.gk = int(prog.meta.get("group_k", 0))           # grouped program: k trajectories over the k members of a group
        self.blob = prog.blob.detach().cpu().numpy()
        self.buf = prog.ops_buffer
        self.ops = prog.ops if member < 0 else prog.meta["member_ops"][member]
        self.ws = np.full(max(prog.ws_floats, 1), np.nan, np.float32)    # this trajectory's block of the global workspace
        self.lds = np.full(prog.traj_floats, np.nan, np.float32)        # NaN poison: an unwritten read shows up
        # kernel start: the whole trajectory region is zeroed once (halo rows and pad channels of the state slot)
        self.lds[:] = 0.0
        self.poison_arena()

    def poison_arena(self):
        """Everything but the x slot is (re)poisoned: ops must write (data rows AND halo rows) before anyone reads."""
        p = self.p
        lo = p.x_off - P2.HALO2 * p.x_stride                     # (compact programs keep the state slot inside the arena)
        hi = p.x_off + (p.horizon + P2.HALO2) * p.x_stride
        keep = self.lds[lo:hi].copy()
        self.lds[:] = np.nan
        if not p.compact:        # (a compact program's state slot is arena memory: nothing of it survives a forward)
            self.lds[lo:hi] = keep

    def load_x(self, x):
        p = self.p
        self.xg = np.asarray(x, np.float32).copy()      # compact programs: the authoritative state lives in global memory (x_out)
        if p.compact:            # the solver step rewrites the slot's halo rows and pad channels too (kernel: `if (S->compact)`)
            self.lds[p.x_off - P2.HALO2 * p.x_stride: p.x_off + (p.horizon + P2.HALO2) * p.x_stride] = 0.0
        for n in range(p.horizon):
            self.lds[p.x_off + n * p.x_stride: p.x_off + n * p.x_stride + p.dim] = x[n]

    def read_slot(self, off, stride, length, chans):
        return np.stack([self.lds[off + n * stride: off + n * stride + chans] for n in range(length)])

    # ------------------------------------------------------------------------------------------ #
    def run_forward(self, emb_row, ctx=None):
        """All ops once; returns the prediction slot (guided programs: see also grad()).  `ctx` (tile, C): the per-sample condition
        features of a batch-tiled MLP program (None: zeros, as for the unconditional forward)."""
        for op in self.ops:
            self.run_op(op, emb_row, ctx)
        p = self.p
        return self.read_slot(p.pred_off, p.pred_stride, p.horizon, p.dim)

    def run_op(self, op, emb_row, ctx=None):
        if True:
            if int(op[P2.W2_KIND]) == P2.KIND2_LOADC:         # context slot <- condition features; halo rows and pad channels zero
                dst, dstr, ln, ch = (int(op[k]) for k in (P2.W2_DST, P2.W2_DST_STRIDE, P2.W2_LOUT, P2.W2_COUT))
                self.lds[dst: dst + (ln + 2 * P2.HALO2) * dstr] = 0.0
                if ctx is not None:
                    for n in range(ln):
                        self.lds[dst + (n + P2.HALO2) * dstr: dst + (n + P2.HALO2) * dstr + ch] = np.asarray(ctx, np.float32)[n]
            elif int(op[P2.W2_KIND]) == P2.KIND2_HEAD:
                self._head(op, np.asarray(emb_row, np.float32))
            elif int(op[P2.W2_KIND]) == P2.KIND2_LOADX:       # slot <- state from global memory; halo rows and pad channels zero
                dst, dstr, ln, ch = (int(op[k]) for k in (P2.W2_DST, P2.W2_DST_STRIDE, P2.W2_LOUT, P2.W2_COUT))
                assert self.p.compact
                self.lds[dst: dst + (ln + 2 * P2.HALO2) * dstr] = 0.0
                for n in range(ln):
                    self.lds[dst + (n + P2.HALO2) * dstr: dst + (n + P2.HALO2) * dstr + ch] = self.xg[n]
            else:
                self._conv(op, np.asarray(emb_row, np.float32))

    def grad(self):
        p = self.p
        return self.read_slot(p.grad_off, p.grad_stride, p.horizon, p.dim)

    def _head(self, op, emb_row):
        """KIND2_HEAD: z = e + W1x . flat(src); gz = w2 * Mish'(z); dst = W1x^T gz; halo rows of dst zeroed."""
        lds = self.lds
        hidden, length, ch = int(op[P2.W2_COUT]), int(op[P2.W2_LOUT]), int(op[P2.W2_LCOLS])
        src, sstr, dst, dstr = (int(op[k]) for k in (P2.W2_RES, P2.W2_RES_STRIDE, P2.W2_DST, P2.W2_DST_STRIDE))
        w1 = self.blob[int(op[P2.W2_BOFF]):int(op[P2.W2_BOFF]) + length * ch * hidden].reshape(length * ch, hidden)
        w2 = self.blob[int(op[P2.W2_GAMMA]):int(op[P2.W2_GAMMA]) + hidden]
        z = emb_row[int(op[P2.W2_EMB]):int(op[P2.W2_EMB]) + hidden].copy()
        for l in range(length):
            for c in range(ch):
                xv = lds[src + (l + P2.HALO2) * sstr + c]
                assert np.isfinite(xv)
                z = z + w1[l * ch + c] * xv
        # the forward value (what the kernel's log_p pass writes to logp_out): w2 . Mish(z) + b2
        self.logp = np.float32((w2 * mish(z)).sum(dtype=np.float32) + self.blob[int(op[P2.W2_GAMMA]) + hidden])
        gz = (w2 * mish_grad(z)).astype(np.float32)
        for l in range(length):
            for c in range(ch):
                acc = np.float32(0.0)
                for j in range(hidden):
                    acc = np.float32(acc + w1[l * ch + c][j] * gz[j])
                lds[dst + (l + P2.HALO2) * dstr + c] = acc
        for r in (0, 1, length + P2.HALO2, length + P2.HALO2 + 1):
            lds[dst + r * dstr: dst + (r + 1) * dstr] = 0.0

    def _conv(self, op, emb_row):
        p, lds = self.p, self.lds
        c_out, l_out, coutp = int(op[P2.W2_COUT]), int(op[P2.W2_LOUT]), int(op[P2.W2_COUTP])
        mode, nt_n, sstride = int(op[P2.W2_MODE]), int(op[P2.W2_NT]), int(op[P2.W2_SSTRIDE])
        l_cols, cstride, ostride = int(op[P2.W2_LCOLS]), int(op[P2.W2_CSTRIDE]), int(op[P2.W2_OSTRIDE])
        flags = int(op[P2.W2_FLAGS])
        lane = np.arange(64)
        if mode == MODE_16X16:
            cols, kstep, lcol, koff, drow = 16, 16, lane & 15, 4 * (lane >> 4), 4 * (lane >> 4)
        else:
            cols, kstep, lcol, koff, drow = 4, 4, lane & 3, 0 * lane, 4 * (lane >> 2)
        stage = p.stage_off
        ksplit = int(op[P2.W2_KSPLIT])
        xg_word = int(op[P2.W2_XG]) if not (flags & (P2.F2_GNBWD | P2.F2_SAVE)) else 0
        gop = bool(self.gk and (xg_word & P2.XG_GOP))        # grouped op: columns = trajectory x position, W2_GMAP maps them to rows
        gsh, grows = (int(op[P2.W2_GMAP]) & 255, int(op[P2.W2_GMAP]) >> 8) if gop else (0, 0)
        l_stage = l_cols if gop else l_out                   # positions per staged K slice
        lds[stage:stage + (ksplit + int(op[P2.W2_KPOST])) * l_stage * sstride] = np.nan      # stale data must not be read

        for item in range(int(op[P2.W2_NITEMS])):             # wave w takes items w, w + 4, ...
            rec = P2.op_item(self.buf, op, item)
            woff, nq, part, col0, ccn = (int(rec[k]) for k in (P2.I2_WOFF, P2.I2_NQ, P2.I2_PART, P2.I2_COL0, P2.I2_CCN))
            tap, cc = int(rec[P2.I2_TAPCC]) & 255, int(rec[P2.I2_TAPCC]) >> 8
            pad, ooff = int(rec[P2.I2_PADOOFF]) & 255, int(rec[P2.I2_PADOOFF]) >> 8
            src, sstr = int(rec[P2.I2_SRCSTR]) & 0xffff, int(rec[P2.I2_SRCSTR]) >> 16
            w4 = self.blob[woff:woff + nq * 256].reshape(nq, 64, 4)
            acc = np.zeros((nt_n, 64, 4), np.float32)         # D fragment: [col tile][lane][4 rows]
            # per-lane operand offset: LINEAR in the tap thanks to the halo; columns past l_cols sit on halo row 0, step 0
            m = np.stack([col0 + nt * cols + lcol for nt in range(nt_n)])
            valid = m < l_cols
            mrow = ((m >> gsh) * grows + (m & ((1 << gsh) - 1)) * cstride) if gop else m * cstride
            row = np.where(valid, mrow - pad + P2.HALO2 + tap, 0)
            assert (row >= 0).all()
            cur = src + row * sstr + koff[None, :] + cc * kstep
            tstep = np.where(valid, sstr, 0) - ccn * kstep
            for q in range(nq):
                a = w4[q]
                for nt in range(nt_n):
                    bmat = np.stack([lds[cur[nt][l]:cur[nt][l] + 4] for l in range(64)])
                    hole = ~np.isfinite(bmat)
                    if hole.any():
                        # pad channel rows of a slot narrower than its 16-row tile (C = 8, 24: never stored; the kernel's LDS
                        # is cleared once per launch and holds finite values ever after): harmless only under all-zero weights
                        if mode == MODE_16X16:
                            dead = (a.reshape(4, 16, 4) == 0).all(axis=1)[:, None, :].repeat(16, axis=1).reshape(64, 4)
                        else:
                            dead = np.broadcast_to((a == 0).all(axis=0)[None, :], (64, 4))
                        assert dead[hole].all(), "B operand read an unwritten LDS word under a non-zero weight"
                        bmat = np.where(hole, np.float32(0), bmat)
                    if mode == MODE_16X16:                     # D[i][j] += sum_k A[i][k] B[k][j]; lane = k*16 + i / k*16 + j
                        d = np.einsum("kim,kjm->ij", a.reshape(4, 16, 4), bmat.reshape(4, 16, 4)).astype(np.float32)
                    else:                                      # 16 blocks of 4x4: row = lane, the 4 columns are lanes 0..3
                        d = (a[:, None, :] * bmat[None, :4, :]).sum(-1).astype(np.float32)
                    for l in range(64):
                        acc[nt][l] += d[drow[l]:drow[l] + 4, lcol[l]]
                cur = cur + kstep                              # cursor walk: chunk, then (every ccn chunks) one tap step
                cc += 1
                if cc == ccn:
                    cc = 0
                    cur = cur + tstep
            for nt in range(nt_n):                             # D fragment -> stage[k slice][output position][row tile + rows]
                for l in range(64):
                    if m[nt][l] < l_cols:
                        n = m[nt][l] * ostride + ooff
                        a0 = stage + part + n * sstride + drow[l]
                        lds[a0:a0 + 4] = acc[nt][l]

        # ---------------- epilogue: thread -> (group, float4 item) ---------------- #
        cg = coutp // P2.GROUPS2
        shift, nk = int(op[P2.W2_CG4_SHIFT]), int(op[P2.W2_NK])
        cg4 = cg // 4
        assert cg4 == 1 << shift
        nv = cg4 * l_out
        dst, dstride, coff = int(op[P2.W2_DST]), int(op[P2.W2_DST_STRIDE]), 0

        def par(word):
            o = int(op[word])
            return self.blob[o:o + coutp]

        bias = emb_row[int(op[P2.W2_BOFF]):int(op[P2.W2_BOFF]) + coutp] if flags & P2.F2_BIAS_EMB else par(P2.W2_BOFF)
        act_id = (flags >> P2.F2_ACT_SHIFT) & 15              # 0: Mish after a GroupNorm, nothing otherwise
        vals = {}
        xg = xg_word                                          # split / grouped programs: this member's lane groups
        g_lo, g_hi = (xg & 255, (xg >> 8) & 255) if xg else (0, P2.GROUPS2)
        if gop:
            # grouped op: the 8 half-waves are (trajectory, lane group of this member) pairs; everything below runs per trajectory on
            # the member's channels, reading stage column t * l_out + pos (channels relative to the member's first one)
            gpm = g_hi - g_lo
            assert gpm * self.gk == P2.GROUPS2
            for t in range(self.gk):
                self._epilogue_rest(op, emb_row, par, bias, act_id, g_lo, g_hi, stage + t * l_out * sstride - g_lo * cg, l_stage,
                                    dst + t * grows * dstride, int(op[P2.W2_RES]) + t * grows * int(op[P2.W2_RES_STRIDE]), {})
            return
        self._epilogue_rest(op, emb_row, par, bias, act_id, g_lo, g_hi, stage, l_out, dst, int(op[P2.W2_RES]), vals)

    def _epilogue_rest(self, op, emb_row, par, bias, act_id, g_lo, g_hi, stage, l_stage, dst, res_off, vals):
        p, lds = self.p, self.lds
        c_out, l_out, coutp = int(op[P2.W2_COUT]), int(op[P2.W2_LOUT]), int(op[P2.W2_COUTP])
        sstride, flags, ksplit = int(op[P2.W2_SSTRIDE]), int(op[P2.W2_FLAGS]), int(op[P2.W2_KSPLIT])
        cg = coutp // P2.GROUPS2
        shift, nk = int(op[P2.W2_CG4_SHIFT]), int(op[P2.W2_NK])
        cg4 = cg // 4
        nv = cg4 * l_out
        dstride, coff = int(op[P2.W2_DST_STRIDE]), 0
        for tid in range(256):
            g, li = tid >> 5, tid & 31
            if not g_lo <= g < g_hi:
                continue
            for k in range(nk):
                i = li + 32 * k
                if i >= nv:
                    continue
                c, pos = g * cg + 4 * (i & (cg4 - 1)), i >> shift
                v = bias[c:c + 4].copy()
                for ks in range(ksplit):
                    a = stage + (ks * l_stage + pos) * sstride + c
                    v = v + lds[a:a + 4]
                vals[(g, pos, c)] = v
        if flags & P2.F2_GNBWD:
            self._epilogue_bwd(op, vals, par, cg, c_out, l_out)
            return
        if (flags & P2.F2_GN) and (flags & P2.F2_COLNORM):
            # per-sample GroupNorm: statistics of one POSITION over the group's channels, two passes (mean, then centred squares)
            gamma, beta = par(P2.W2_GAMMA), par(P2.W2_BETA)
            inv_cnt = np.int32(op[P2.W2_INV_CNT]).view(np.float32)
            for g in range(P2.GROUPS2):
                for pos in range(l_out):
                    keys = [kk for kk in vals if kk[0] == g and kk[1] == pos]
                    allv = np.concatenate([vals[kk] for kk in keys])
                    mean = np.float32(allv.sum(dtype=np.float32) * inv_cnt)
                    creal4 = int(op[P2.W2_CGREAL4])              # narrower real groups: the zero pad channels stay out of the variance
                    real = [kk for kk in keys if not creal4 or (kk[2] - g * cg) // 4 < creal4]
                    dlt = (np.concatenate([vals[kk] for kk in real]) - mean).astype(np.float32)
                    rstd = np.float32(1.0) / np.sqrt(np.float32((dlt * dlt).sum(dtype=np.float32) * inv_cnt) + np.float32(GN_EPS))
                    for kk in keys:
                        c = kk[2]
                        y = ((vals[kk] - mean) * rstd).astype(np.float32) * gamma[c:c + 4] + beta[c:c + 4]
                        vals[kk] = activation(y, act_id - 1) if act_id else mish(y)
        elif flags & P2.F2_GN:
            gamma, beta = par(P2.W2_GAMMA), par(P2.W2_BETA)
            inv_cnt = np.int32(op[P2.W2_INV_CNT]).view(np.float32)
            for g in range(g_lo, g_hi):
                keys = [kk for kk in vals if kk[0] == g]
                allv = np.concatenate([vals[kk] for kk in keys])
                # single pass, shifted by the group's first element (lane 0 of the half-wave, component 0): one cross-lane
                # reduction round instead of two, without the cancellation of a raw E[x^2] - E[x]^2
                shift = vals[(g, 0, g * cg)][0]
                dlt = (allv - shift).astype(np.float32)
                m1 = np.float32(dlt.sum(dtype=np.float32) * inv_cnt)
                var = np.float32((dlt * dlt).sum(dtype=np.float32) * inv_cnt - m1 * m1)
                mean = np.float32(shift + m1)
                rstd = np.float32(1.0) / np.sqrt(var + np.float32(GN_EPS))
                if flags & P2.F2_SAVE:
                    lds[int(op[P2.W2_STATS]) + g] = rstd
                for kk in keys:
                    c = kk[2]
                    xh = ((vals[kk] - mean) * rstd).astype(np.float32)
                    if (flags & P2.F2_SAVE) and c < c_out:  # save slot: no halo, position-major; pad lane groups save nothing
                        a = int(op[P2.W2_SAVE]) + kk[1] * int(op[P2.W2_SAVE_STRIDE]) + c
                        (self.ws if flags & P2.F2_SAVE_GLOBAL else lds)[a:a + 4] = xh
                    y = xh * gamma[c:c + 4] + beta[c:c + 4]
                    vals[kk] = activation(y, act_id - 1) if act_id else mish(y)
        elif act_id:
            for kk in vals:
                vals[kk] = activation(vals[kk], act_id - 1)
        kpost = int(op[P2.W2_KPOST])
        for (g, pos, c), v in vals.items():
            if flags & P2.F2_FILM:                            # [scale | bias], pad32(C) apart: y <- scale * y + bias (two roundings)
                e0 = int(op[P2.W2_EMB]) + c
                v = (v * emb_row[e0:e0 + 4]).astype(np.float32) + emb_row[e0 + coutp:e0 + coutp + 4]
            elif flags & P2.F2_EMB:
                e0 = int(op[P2.W2_EMB]) + c
                v = v + emb_row[e0:e0 + 4]
            if kpost:                                         # extra conv(s) added after the norm: bias + their staged partials
                pv = par(P2.W2_PBIAS)[c:c + 4].copy()
                for ks in range(ksplit, ksplit + kpost):
                    a = stage + (ks * l_stage + pos) * sstride + c
                    pv = pv + lds[a:a + 4]
                v = v + pv
            if flags & P2.F2_RES:
                a = res_off + (pos + P2.HALO2) * int(op[P2.W2_RES_STRIDE]) + c
                v = v + lds[a:a + 4]
            if flags & P2.F2_OUT_DIV:
                v = (v / np.int32(op[P2.W2_ODIV]).view(np.float32)).astype(np.float32)
            for j in range(4):
                if c + j < c_out:
                    assert np.isfinite(v[j]), "NaN reached a destination slot"
                    lds[dst + (pos + P2.HALO2) * dstride + coff + c + j] = v[j]
        for r in (0, 1, l_out + P2.HALO2, l_out + P2.HALO2 + 1):        # wave w rewrites halo row w of the destination
            lds[dst + r * dstride: dst + (r + 1) * dstride] = 0.0
        xgw = int(op[P2.W2_XG]) if not (flags & (P2.F2_GNBWD | P2.F2_SAVE)) else 0
        if self.gk and (xgw & P2.XG_TRAJ):
            # an ordinary op that wrote this member's trajectory into a group slot: the halo rows of the OTHER sub-slots too (the exchange
            # brings their data rows only)
            grows = int(op[P2.W2_GMAP]) >> 8
            base = dst - self.member * grows * dstride
            for t in range(self.gk):
                for r in (0, 1, l_out + P2.HALO2, l_out + P2.HALO2 + 1):
                    lds[base + (t * grows + r) * dstride: base + (t * grows + r + 1) * dstride] = 0.0

    def _epilogue_bwd(self, op, vals, par, cg, c_out, l_out):
        """F2_GNBWD: v (+ residual slot) -> [dst2]; g_xhat = v Mish'(gamma x_hat + beta) gamma; dst = rstd (g_xhat - mean(g_xhat) -
        x_hat mean(g_xhat x_hat)), means over the group."""
        lds = self.lds
        flags = int(op[P2.W2_FLAGS])
        gamma, beta = par(P2.W2_GAMMA), par(P2.W2_BETA)
        inv_cnt = np.int32(op[P2.W2_INV_CNT]).view(np.float32)
        dst, dstride = int(op[P2.W2_DST]), int(op[P2.W2_DST_STRIDE])
        gx, xhs = {}, {}
        for (g, pos, c), v in vals.items():
            if c >= c_out:                                  # pad lane group (fewer than 8 x 4 channels): never stored, skip
                continue
            if flags & P2.F2_RES:
                a = int(op[P2.W2_RES]) + (pos + P2.HALO2) * int(op[P2.W2_RES_STRIDE]) + c
                assert np.isfinite(lds[a:a + 4]).all()
                v = v + lds[a:a + 4]
            if flags & P2.F2_DUAL:
                a = int(op[P2.W2_DST2]) + (pos + P2.HALO2) * int(op[P2.W2_DST2_STRIDE]) + c
                lds[a:a + 4] = v
            a = int(op[P2.W2_SAVE]) + pos * int(op[P2.W2_SAVE_STRIDE]) + c
            xh = (self.ws if flags & P2.F2_SAVE_GLOBAL else lds)[a:a + 4].copy() if c < c_out else np.zeros(4, np.float32)
            assert np.isfinite(xh).all(), "backward read an unsaved x_hat"
            d = mish_grad(xh * gamma[c:c + 4] + beta[c:c + 4])
            gx[(g, pos, c)] = (v * d * gamma[c:c + 4]).astype(np.float32)
            xhs[(g, pos, c)] = xh
        for g in range(P2.GROUPS2):
            keys = [kk for kk in gx if kk[0] == g]
            if not keys:
                continue
            s1 = np.float32(sum(np.float32(gx[kk].sum(dtype=np.float32)) for kk in keys))
            s2 = np.float32(sum(np.float32((gx[kk] * xhs[kk]).sum(dtype=np.float32)) for kk in keys))
            m1, m2 = np.float32(s1 * inv_cnt), np.float32(s2 * inv_cnt)
            rstd = lds[int(op[P2.W2_STATS]) + g]
            assert np.isfinite(rstd)
            for kk in keys:
                _, pos, c = kk
                gu = ((gx[kk] - m1 - xhs[kk] * m2) * rstd).astype(np.float32)
                for j in range(4):
                    if c + j < c_out:
                        assert np.isfinite(gu[j])
                        lds[dst + (pos + P2.HALO2) * dstride + c + j] = gu[j]
        for r in (0, 1, l_out + P2.HALO2, l_out + P2.HALO2 + 1):
            lds[dst + r * dstride: dst + (r + 1) * dstride] = 0.0
            if flags & P2.F2_DUAL:
                d2, d2s = int(op[P2.W2_DST2]), int(op[P2.W2_DST2_STRIDE])
                lds[d2 + r * d2s: d2 + (r + 1) * d2s] = 0.0


def chi_film_rows(prog: P2.Program2, net, t, cond) -> np.ndarray:
    """FiLM table rows of a ChiUNet1d program (what runtime2.chi_film_table computes on the device): one row per (timestep, sample)
    pair given elementwise -- t (n,), cond (n, To, obs) -> (n, n_emb)."""
    import torch
    import torch.nn.functional as F
    f = prog.meta["chi_film"]
    with torch.no_grad():
        te = F.mish(net.map_emb(net.map_noise(t)))
        ce = F.mish(net.global_cond_encoder(torch.flatten(cond, 1)))
        rows = te @ f["w_t"].cpu().t() + ce @ f["w_c"].cpu().t() + f["bias"].cpu()
    return rows.numpy().astype(np.float32)


def mlp_rows(prog: P2.Program2, net, t) -> np.ndarray:
    """Per-step table rows of a batch-tiled MLP program (what runtime2.mlp_table computes on the device): bias + W_src feat_src(t) for
    src in {"temb": map_noise(t), "tfeat": the net's time MLP over it, "t": the raw timestep}; t (n,) -> (n, n_emb)."""
    import torch
    spec = prog.meta["mlp"]
    r = spec["rows"]
    with torch.no_grad():
        temb = net.map_noise(t)
        out = r["bias"].cpu()[None, :].repeat(t.shape[0], 1)
        if "temb" in r:
            out = out + temb @ r["temb"].cpu().t()
        if "tfeat" in r:
            tf = net.time_mlp(temb) if spec["kind"] == "dql" else net.t_layer(temb)
            out = out + tf @ r["tfeat"].cpu().t()
        if "t" in r:
            out = out + t.to(torch.float32)[:, None] @ r["t"].cpu().t()
    return out.numpy().astype(np.float32)


def _xg(op) -> int:
    """The op's cut / exchange word: W2_XG aliases W2_DST2, which ops with backward extras (the classifier's part of a split / grouped
    GUIDED program) use under its own name -- they are never cut (kernel: run_op)."""
    return 0 if int(op[P2.W2_FLAGS]) & (P2.F2_GNBWD | P2.F2_SAVE | P2.F2_DUAL) else int(op[P2.W2_XG])


def run_forward_split(sims, emb_row):
    """One forward of a SPLIT program: `sims[m]` = LaneSim2(prog, member=m), all loaded with the same state.  The members step through
    the op list together; after an op that is cut over the members (W2_XG) every member receives the channels it did not compute from
    the member that did -- the all-gather the kernel performs through global memory.  Returns member 0's prediction slot (all equal)."""
    p = sims[0].p
    k = len(sims)
    for i in range(len(sims[0].ops)):
        for s in sims:
            s.run_op(s.ops[i], emb_row)
        xgs = [_xg(s.ops[i]) for s in sims]
        if not any(xgs):
            continue
        assert all(xgs), "an op is split for every member or for none"
        op = sims[0].ops[i]
        c_out, l_out, coutp = int(op[P2.W2_COUT]), int(op[P2.W2_LOUT]), int(op[P2.W2_COUTP])
        cg = coutp // P2.GROUPS2
        owner = {}
        for m, xg in enumerate(xgs):
            for g in range(xg & 255, (xg >> 8) & 255):
                owner.setdefault(g, m)
        assert sorted(owner) == list(range(min(P2.GROUPS2, -(-c_out // cg)))) or len(owner) == P2.GROUPS2, owner
        for m, s in enumerate(sims):
            lo, hi = xgs[m] & 255, (xgs[m] >> 8) & 255
            dst, dstr = int(s.ops[i][P2.W2_DST]), int(s.ops[i][P2.W2_DST_STRIDE])
            for g, src_m in owner.items():
                if lo <= g < hi:
                    continue
                src = sims[src_m]
                sd, sds = int(src.ops[i][P2.W2_DST]), int(src.ops[i][P2.W2_DST_STRIDE])
                for pos in range(l_out):
                    for c in range(g * cg, min((g + 1) * cg, c_out)):
                        s.lds[dst + (pos + P2.HALO2) * dstr + c] = src.lds[sd + (pos + P2.HALO2) * sds + c]
    return sims[0].read_slot(p.pred_off, p.pred_stride, p.horizon, p.dim)


def run_forward_group(sims, emb_row):
    """One forward of a GROUPED program (engine/program2.py:compile_janner2_group): `sims[m]` = LaneSim2(prog, member=m), member m loaded
    with trajectory m of the group.  The members step through the op list together.  After a grouped op (XG_GOP) every member holds its
    lane groups of ALL k trajectories and receives the other lane groups from the members that computed them; after an ordinary op that
    wrote the member's own trajectory into a group slot (XG_TRAJ) every member receives the other members' whole trajectories -- the two
    all-gathers the kernel performs through the group's tile in L2.  Returns the k prediction slots (member m: trajectory m)."""
    p = sims[0].p
    k = len(sims)
    assert k == p.meta["group_k"]
    for i in range(len(sims[0].ops)):
        for s in sims:
            s.run_op(s.ops[i], emb_row)
        xgs = [_xg(s.ops[i]) for s in sims]
        if not any(x & P2.XG_XCHG for x in xgs):
            continue
        assert all(x & P2.XG_XCHG for x in xgs), "an op is exchanged by every member or by none"
        op = sims[0].ops[i]
        c_out, l_out, coutp = int(op[P2.W2_COUT]), int(op[P2.W2_LOUT]), int(op[P2.W2_COUTP])
        cg = coutp // P2.GROUPS2
        grows = int(op[P2.W2_GMAP]) >> 8
        assert grows == l_out + 2 * P2.HALO2
        if xgs[0] & P2.XG_GOP:
            owner = {}
            for m, xg in enumerate(xgs):
                assert xg & P2.XG_GOP
                for g in range(xg & 255, (xg >> 8) & 255):
                    assert g not in owner
                    owner[g] = m
            assert sorted(owner) == list(range(P2.GROUPS2))
            for m, s in enumerate(sims):
                dst, dstr = int(s.ops[i][P2.W2_DST]), int(s.ops[i][P2.W2_DST_STRIDE])
                for g, src_m in owner.items():
                    if src_m == m:
                        continue
                    src = sims[src_m]
                    sd = int(src.ops[i][P2.W2_DST])
                    for t in range(k):
                        for pos in range(l_out):
                            a = (t * grows + pos + P2.HALO2) * dstr
                            s.lds[dst + a + g * cg: dst + a + min((g + 1) * cg, c_out)] = src.lds[sd + a + g * cg: sd + a + min((g + 1) * cg, c_out)]
        else:
            assert all(x & P2.XG_TRAJ for x in xgs)
            # every member's W2_DST points at ITS sub-slot: the group slot starts `m` sub-slots before it
            dstr = int(op[P2.W2_DST_STRIDE])
            bases = [int(s.ops[i][P2.W2_DST]) - m * grows * dstr for m, s in enumerate(sims)]
            assert len(set(bases)) == 1
            for m, s in enumerate(sims):
                for t in range(k):
                    if t == m:
                        continue
                    for pos in range(l_out):             # data rows only, real channels only: the receiving member zeroes the halo rows
                        lo = bases[0] + (t * grows + pos + P2.HALO2) * dstr      # of every sub-slot itself (kernel: halo waves)
                        s.lds[lo:lo + c_out] = sims[t].lds[lo:lo + c_out]
    return [s.read_slot(p.pred_off, p.pred_stride, p.horizon, p.dim) for s in sims]
