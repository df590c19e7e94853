"""Lane-level numpy model of ``csrc/cdx_unet1d.hip`` -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Interprets the SAME (ops, blob, LDS plan) the kernel receives, with the kernel's exact index arithmetic:
work items (ct, ks) per wave, chunk enumeration (source, tap, cc), the 64-lane A/B operand fetch of
``v_mfma_f32_16x16x4_f32`` (A: i = lane & 15, k = lane >> 4; B: k = lane >> 4, j = lane & 15; D: col = lane & 15,
row = 4 * (lane >> 4) + r), the split-K scratch layout and the GroupNorm/Mish/FiLM/residual epilogue.  It exists so
that the weight packing, slot allocation and addressing are proven against the reference fixtures on CPU before
a GPU minute is spent; it is also the executable specification the HIP code was transcribed from.
"""
import numpy as np

from cleandiffuser_amd.engine import program as P


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


class LaneSim:
    def __init__(self, prog: P.Program):
        self.p = prog
        self.blob = prog.blob.detach().cpu().numpy()
        self.lds = np.full(prog.lds_floats, np.nan, np.float32)   # NaN-poison: reading an unwritten word shows up

    # ------------------------------------------------------------------------------------------ #
    def load_x(self, x, cond_rows=None):                  # x [H][D] -> padded channel-last slot, halo/pad zeroed
        p = self.p
        rows = p.horizon + 2 * P.HALO
        self.lds[p.x_off:p.x_off + rows * p.x_stride] = 0.0
        self.lds[p.zero_off:p.zero_off + p.zero_floats] = 0.0   # kernel-lifetime slots + the shared zero row
        if p.tile and cond_rows is not None:              # tile programs: per-sample condition features -> context slot
            for n in range(p.horizon):
                base = p.cond_slot_off + (n + P.HALO) * p.cond_slot_stride + p.cond_coff
                self.lds[base:base + p.cond_dim] = cond_rows[n]
        for n in range(p.horizon):
            base = p.x_off + (n + P.HALO) * p.x_stride
            self.lds[base:base + p.dim] = x[n]

    def read_slot(self, off, stride, length, chans):
        out = np.empty((length, chans), np.float32)
        for n in range(length):
            base = off + (n + P.HALO) * stride
            out[n] = self.lds[base:base + chans]
        return out

    # ------------------------------------------------------------------------------------------ #
    def run_forward(self, temb_row, cond_row=None, branch=0):
        for op in self.p.ops:
            kind = op[P.W_KIND]
            if kind == P.OP_LOAD_TEMB:
                n, dst = op[P.L_NIN], op[P.L_DST]
                v = np.asarray(temb_row, np.float32).copy()
                if cond_row is not None and not self.p.tile and not self.p.cond_dim:
                    v = v + cond_row
                self.lds[dst:dst + n] = v
            elif kind == P.OP_LINEAR:
                self._linear(op)
            elif kind == P.OP_LOAD_COND:
                n, dst = op[P.L_NIN], op[P.L_DST]
                self.lds[dst:dst + n] = cond_row if cond_row is not None else 0.0
            elif kind == P.OP_FILL:
                n, rows, src, dst, sstr, coff = (op[P.L_NIN], op[P.L_NOUT], op[P.L_SRC], op[P.L_DST], op[P.L_WOFF],
                                                 op[P.L_BOFF])
                for r in range(rows):
                    base = dst + (r + P.HALO) * sstr + coff
                    self.lds[base:base + n] = self.lds[src:src + n]
            elif kind == P.OP_FLATTEN:
                c, n, src, dst, sstr = op[P.L_NIN], op[P.L_NOUT], op[P.L_SRC], op[P.L_DST], op[P.L_WOFF]
                self.lds[dst:dst + c * n] = self.read_slot(src, sstr, n, c).T.reshape(-1)   # (C, L) flatten order
            else:
                self._conv(op, branch)
        p = self.p
        if p.out_vec_len:
            return self.lds[p.out_vec_off:p.out_vec_off + p.out_vec_len].copy()
        return self.read_slot(p.pred_off + branch * p.pred_branch_floats, p.pred_stride, p.horizon, p.dim)

    def _linear(self, op):
        n_in, n_out = op[P.L_NIN], op[P.L_NOUT]
        w = self.blob[op[P.L_WOFF]:op[P.L_WOFF] + n_in * n_out].reshape(n_in, n_out)
        bvec = self.blob[op[P.L_BOFF]:op[P.L_BOFF] + n_out]
        x = self.lds[op[P.L_SRC]:op[P.L_SRC] + n_in]
        acc = bvec.copy()
        for i in range(n_in):                              # same i-order as the kernel's per-thread loop
            acc = acc + w[i] * x[i]
        if op[P.L_FLAGS] & P.F_RAW_COPY:
            self.lds[op[P.L_DST2]:op[P.L_DST2] + n_out] = acc
        if op[P.L_FLAGS] & P.F_POST_MISH:
            acc = mish(acc)
        self.lds[op[P.L_DST]:op[P.L_DST] + n_out] = acc

    def _row(self, op, pos, tap):
        """LDS row (halo included) that output position `pos` reads for `tap`, or -1 if it contributes zero."""
        if op[P.W_TRANSPOSED]:
            num = pos + op[P.W_CPAD] - tap
            if num % op[P.W_CSTRIDE] != 0:
                return -1
            q = num // op[P.W_CSTRIDE]
        else:
            q = pos * op[P.W_CSTRIDE] + tap - op[P.W_CPAD]
        return q + P.HALO if 0 <= q < op[P.W_LIN] else -1

    def _conv(self, op, branch):
        p, lds = self.p, self.lds
        c_out, c16, l_out = op[P.W_COUT], op[P.W_COUT16], op[P.W_LOUT]
        n_ct, taps = c16 // 16, op[P.W_TAPS]
        ksplit, n_chunks = op[P.W_KSPLIT], op[P.W_NCHUNKS]
        ca, cb = op[P.W_CA_CHUNKS], op[P.W_CB_CHUNKS]
        assert n_chunks == taps * (ca + cb)
        flags = op[P.W_FLAGS]
        dst = op[P.W_DST] + (branch * p.pred_branch_floats if flags & P.F_DST_PRED else 0)
        dstride, drows = op[P.W_DST_STRIDE], op[P.W_DST_ROWS]
        if not flags & (P.F_ACCUM | P.F_KEEP_DST):
            lds[dst:dst + drows * dstride] = 0.0           # kernel zeroes the whole slot before the K loop
        sstride = c16 + 4
        scratch = p.scratch_off
        lane = np.arange(64)
        mode = op[P.W_MODE]
        buf = p.ops_buffer
        for item in range(op[P.W_NITEMS]):                 # wave w takes items w, w+8, ...
            rec = buf[op[P.W_ITEMS] + item * P.ITEM_WORDS: op[P.W_ITEMS] + (item + 1) * P.ITEM_WORDS]
            woff, part, nq = int(rec[P.I_WOFF]), int(rec[P.I_PART]), int(rec[P.I_NQ])
            onb, tap, cc = int(rec[P.I_ONB]), int(rec[P.I_TAP]), int(rec[P.I_CC])
            w4 = self.blob[woff:woff + nq * 256].reshape(nq, 64, 4)
            if mode == P.MODE_16X16:
                n_nt, cols, kstep = (l_out + 15) // 16, 16, 16
                li, lk = lane & 15, lane >> 4
            else:
                n_nt, cols, kstep = (l_out + 3) // 4, 4, 4
                li, lk = lane & 3, lane >> 2                # li = column j, lk = block (4 output rows each)
            acc = np.zeros((n_nt, 64, cols), np.float32)   # [col tile][row within row tile][col]
            for q in range(nq):
                src, sstr, ccn = ((op[P.W_SRCB], op[P.W_SRCB_STRIDE], cb) if onb
                                  else (op[P.W_SRCA], op[P.W_SRCA_STRIDE], ca))
                a = w4[q]                                   # [lane][m]
                for nt in range(n_nt):
                    bmat = np.zeros((64, 4), np.float32)
                    for l in range(64):
                        pos = nt * cols + li[l]
                        row = self._row(op, pos, tap) if pos < l_out else -1
                        base = src + row * sstr if row >= 0 else p.zrow_off   # non-contributing taps: shared zero row
                        addr = base + cc * kstep + (4 * lk[l] if mode == P.MODE_16X16 else 0)
                        bmat[l] = lds[addr:addr + 4]
                    for m in range(4):
                        if mode == P.MODE_16X16:            # D[i][j] += sum_k A[i][k] B[k][j]
                            amat = np.zeros((16, 4), np.float32)
                            kmat = np.zeros((4, 16), np.float32)
                            amat[lane & 15, lane >> 4] = a[:, m]
                            kmat[lane >> 4, lane & 15] = bmat[:, m]
                            acc[nt][:16] += amat @ kmat
                        else:                               # 16 blocks: D_b[i][j] += A_b[i] * B_b[j], row = 4b + i
                            avec = a[:, m]                  # lane = 4b + i  -> row
                            bvec = bmat[:4, m]              # every block reads the same 4 columns (lanes 0..3)
                            acc[nt] += np.outer(avec, bvec)
                if q + 1 < nq:                              # cursor walk (source, tap, chunk)
                    cc += 1
                    if cc == ccn:
                        cc = 0
                        tap += 1
                        if tap == taps:
                            tap, onb = 0, 1
            rows = 16 if mode == P.MODE_16X16 else 64
            for nt in range(n_nt):
                for j in range(cols):
                    n = nt * cols + j
                    if n < l_out:
                        base = scratch + part + n * sstride
                        lds[base:base + rows] = acc[nt][:rows, j]
        # ---------------- epilogue ---------------- #
        bias = self.blob[op[P.W_BOFF]:op[P.W_BOFF] + c_out]
        v = np.zeros((l_out, c_out), np.float32)
        for n in range(l_out):
            row = bias.copy()
            for ks in range(ksplit):
                base = scratch + (ks * l_out + n) * sstride
                row = row + lds[base:base + c_out]
            v[n] = row
        norm, act_id = op[P.W_NORM], op[P.W_ACT]
        if norm != P.NORM_NONE:
            g = op[P.W_GROUPS]
            cg = c_out // g
            gamma = self.blob[op[P.W_GAMMA]:op[P.W_GAMMA] + c_out]
            beta = self.blob[op[P.W_BETA]:op[P.W_BETA] + c_out]
            for gi in range(g):
                cols = slice(gi * cg, (gi + 1) * cg)
                # slot-group: statistics over (positions x channels of the group); column: per position (sample)
                axes = None if norm == P.NORM_SLOT_GROUP else 1
                blk = v[:, cols]
                cnt = np.float32(blk.size if axes is None else cg)
                mean = (blk.sum(axis=axes, keepdims=True, dtype=np.float32) / cnt).astype(np.float32)
                dev = blk - mean
                var = ((dev * dev).sum(axis=axes, keepdims=True, dtype=np.float32) / cnt).astype(np.float32)
                rstd = (np.float32(1.0) / np.sqrt(var + np.float32(P.GN_EPS))).astype(np.float32)
                v[:, cols] = dev * rstd * gamma[cols] + beta[cols]
        v = activation(v, act_id)
        if flags & P.F_ADD_EMB:
            v = v + lds[op[P.W_EMB]:op[P.W_EMB] + c_out][None, :]
        if flags & P.F_FILM:
            e = lds[op[P.W_EMB]:op[P.W_EMB] + 2 * c_out]
            v = v * e[None, :c_out] + e[None, c_out:]
        if flags & P.F_ADD_RES:
            v = v + self.read_slot(op[P.W_RES], op[P.W_RES_STRIDE], l_out, c_out)
        if flags & P.F_SCALE:
            v = v * np.int32(op[P.W_SCALE]).view(np.float32)
        coff = op[P.W_DST_COFF]
        for n in range(l_out):
            base = dst + (n + P.HALO) * dstride + coff
            if flags & P.F_ACCUM:
                lds[base:base + c_out] += v[n]
            else:
                lds[base:base + c_out] = v[n]
        if flags & P.F_KEEP_DST:                           # partial write into a longer-lived slot: check what was written
            for n in range(l_out):
                base = dst + (n + P.HALO) * dstride + coff
                assert np.isfinite(lds[base:base + c_out]).all(), "NaN reached a destination slot"
        else:
            assert np.isfinite(lds[dst:dst + drows * dstride]).all(), "NaN reached a destination slot"
