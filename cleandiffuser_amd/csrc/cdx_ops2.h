// cdx_ops2.h -- word layout of the v2 U-Net program (csrc/cdx_unet2.hip): op descriptors of CDX2_HDR_WORDS header words followed
// by one inline CDX2_ITEM_WORDS-word work item per wave of the workgroup shape the program was compiled for (4 or 8 waves).
// MUST mirror cleandiffuser_amd/engine/program2.py (tests/test_abi_contract.py parses this file and compares).
#pragma once

/* shared with engine/consts.py (MODE_*, GN_EPS): MFMA shape of a conv op's records, GroupNorm epsilon */
#define CDX_MODE_16X16 0
#define CDX_MODE_4X4 1
#define CDX_GN_EPS 1e-5f

#define CDX2_HDR_WORDS 32   /* 25 descriptor words, padded; == CDX2_W2_ITEM0 */
#define CDX2_ITEM_WORDS 8
#define CDX2_NW2 4         /* waves per workgroup of the default shape (one per SIMD) */
#define CDX2_NW2_MAX 8     /* ... and of the two-waves-per-SIMD shape */
#define CDX2_RING2 16      /* 1-KiB weight records in flight per wave, 4-wave shape */
#define CDX2_RING2_NW8 8   /* ... 8-wave shape */
#define CDX2_GROUPS2 8    /* GroupNorm groups: 32 lanes per group in the epilogue */
#define CDX2_MAX_NK2 4    /* float4 items one lane may own in the epilogue */
#define CDX2_HALO2 2       /* zero rows on either side of an activation slot */

#define CDX2_W2_KIND 0
#define CDX2_W2_FLAGS 1
#define CDX2_W2_COUT 2
#define CDX2_W2_LOUT 3
#define CDX2_W2_LCOLS 4      /* tile columns: l_out, or l_in for the per-parity halves of a transposed conv */
#define CDX2_W2_CSTRIDE 5
#define CDX2_W2_OSTRIDE 6    /* output position = column * OSTRIDE + item offset (2 for a transposed conv) */
#define CDX2_W2_MODE 7
#define CDX2_W2_NT 8
#define CDX2_W2_NITEMS 9
#define CDX2_W2_ITEMS 10     /* word offset of items n_waves.. (the first n_waves are inline at W2_ITEM0) */
#define CDX2_W2_DST 11
#define CDX2_W2_DST_STRIDE 12
#define CDX2_W2_SSTRIDE 13
#define CDX2_W2_KSPLIT 14
#define CDX2_W2_BOFF 15
#define CDX2_W2_GAMMA 16
#define CDX2_W2_BETA 17
#define CDX2_W2_EMB 18
#define CDX2_W2_RES 19
#define CDX2_W2_RES_STRIDE 20
#define CDX2_W2_CG4_SHIFT 21
#define CDX2_W2_INV_CNT 22
#define CDX2_W2_NK 23
#define CDX2_W2_COUTP 24
#define CDX2_W2_SAVE 25        /* slot (no halo) of the normalised pre-affine values a GroupNorm layer keeps for its backward */
#define CDX2_W2_SAVE_STRIDE 26
#define CDX2_W2_STATS 27       /* 8 floats: rstd per GroupNorm group */
#define CDX2_W2_DST2 28        /* F2_DUAL: slot of the value before the backward epilogue */
#define CDX2_W2_DST2_STRIDE 29
#define CDX2_W2_KPOST 30       /* partial tiles after the first KSPLIT that are added AFTER norm / activation (fused 1x1 skip conv) */
#define CDX2_W2_PBIAS 31       /* blob offset of their bias */
#define CDX2_KIND2_CONV 0
#define CDX2_KIND2_HEAD 1      /* classifier head, forward + backward (words: COUT hidden, LOUT/LCOLS positions/channels, RES src,
                                * DST gradient slot, BOFF W1x [l][c][hidden], GAMMA w2, EMB table offset) */
#define CDX2_KIND2_LOADX 2     /* compact guided programs: slot DST (COUT channels, LOUT positions) <- the trajectory's state x_t, read
                                * back from the launch's x_out (where compact programs keep it); halo rows and pad channels zeroed */
#define CDX2_W2_ITEM0 32

#define CDX2_I2_WOFF 0
#define CDX2_I2_NQ 1
#define CDX2_I2_TAPCC 2      /* start cursor: tap | chunk << 8 */
#define CDX2_I2_PART 3
#define CDX2_I2_COL0 4
#define CDX2_I2_PADOOFF 5    /* conv padding | output-position offset << 8 */
#define CDX2_I2_SRCSTR 6     /* source slot: float offset | row stride << 16 */
#define CDX2_I2_CCN 7        /* chunks per tap */

#define CDX2_F2_GN 1
#define CDX2_F2_EMB 2
#define CDX2_F2_RES 4
#define CDX2_F2_PRED 8
#define CDX2_F2_SAVE 16      /* forward GroupNorm op also stores x_hat and rstd */
#define CDX2_F2_GNBWD 32     /* epilogue = backward of (GroupNorm -> Mish) of the layer named by W2_SAVE / W2_STATS */
#define CDX2_F2_DUAL 64      /* ... and the value before that backward goes to W2_DST2 */
#define CDX2_F2_SAVE_GLOBAL 128   /* W2_SAVE is a float offset inside the trajectory's block of cdx_unet2_launch.ws, not an LDS slot */
#define CDX2_F2_FILM 256     /* with F2_EMB: the table row holds [scale | bias] (2 x W2_COUTP floats) at W2_EMB: y <- scale * y + bias */
/* batch-tiled MLP programs (MLP kernel instantiation): */
#define CDX2_F2_COLNORM 512  /* with F2_GN: statistics per POSITION over the group's channels (per-sample GroupNorm) */
#define CDX2_F2_BIAS_EMB 1024   /* the bias vector is read from the per-step table row at W2_BOFF */
#define CDX2_F2_OUT_DIV 2048    /* the stored value is divided by the float in W2_ODIV */
#define CDX2_F2_ACT_SHIFT 12    /* flags bits 12-15: activation id + 1 (CDX_ACT_* of include/cdx.h), 0 = Mish after a GroupNorm, else none */
#define CDX2_W2_ODIV 26         /* forward ops: alias of W2_SAVE_STRIDE */
#define CDX2_W2_XG 28           /* forward ops of a split program: alias of W2_DST2: lane groups [lo, hi) of this member as lo | hi << 8 | 1 << 16; 0 = not cut */
#define CDX2_XG_XCHG 65536    /* W2_XG: the members exchange this op's output afterwards */
#define CDX2_XG_GOP 131072     /* ... a GROUPED op: tile columns = (trajectory of the group) x position, W2_GMAP maps a column to its input row */
#define CDX2_XG_TRAJ 262144    /* ... an ordinary op that wrote the member's trajectory into a group slot: whole trajectories are gathered */
#define CDX2_W2_GMAP 25         /* forward ops of a grouped program: alias of W2_SAVE: log2(positions per trajectory) | rows per sub-slot << 8 */
#define CDX2_W2_CGREAL4 29      /* F2_COLNORM ops: alias of W2_DST2_STRIDE: float4 items per lane group that hold real channels (0 = all) */
#define CDX2_KIND2_LOADC 3      /* context slot <- the launch's per-sample condition features (zeros: unconditional forward / no condition) */
