// cdx_ops.h -- word layout of one layer descriptor (CDX_OP_WORDS int32 each).
// MUST mirror cleandiffuser_amd/engine/program.py (tests/test_abi_contract.py parses this file and compares).
#pragma once

#define CDX_OP_LOAD_TEMB 0
#define CDX_OP_LINEAR 1
#define CDX_OP_CONV 2
#define CDX_OP_FLATTEN 3
#define CDX_OP_FILL 4
#define CDX_OP_LOAD_COND 5

#define CDX_W_KIND 0
// ---- conv ----
#define CDX_W_COUT 1
#define CDX_W_COUT16 2
#define CDX_W_LOUT 3
#define CDX_W_TAPS 4
#define CDX_W_CSTRIDE 5
#define CDX_W_CPAD 6
#define CDX_W_TRANSPOSED 7
#define CDX_W_SRCA 8
#define CDX_W_SRCA_STRIDE 9
#define CDX_W_CA_CHUNKS 10
#define CDX_W_SRCB 11
#define CDX_W_SRCB_STRIDE 12
#define CDX_W_CB_CHUNKS 13
#define CDX_W_DST 14
#define CDX_W_DST_STRIDE 15
#define CDX_W_DST_ROWS 16
#define CDX_W_WOFF 17
#define CDX_W_BOFF 18
#define CDX_W_FLAGS 19
#define CDX_W_GROUPS 20
#define CDX_W_GAMMA 21
#define CDX_W_BETA 22
#define CDX_W_EMB 23
#define CDX_W_RES 24
#define CDX_W_RES_STRIDE 25
#define CDX_W_KSPLIT 26
#define CDX_W_NCHUNKS 27
#define CDX_W_LIN 28
#define CDX_W_MODE 29
#define CDX_W_ITEMS 30
#define CDX_W_NITEMS 31
#define CDX_W_INV_CNT 32
#define CDX_W_CG 33
#define CDX_W_CG_SHIFT 34
#define CDX_W_INV_COUT 35
#define CDX_W_ACT 36
#define CDX_W_NORM 37
#define CDX_W_SCALE 38
#define CDX_W_DST_COFF 39
// ---- linear / load_temb ----
#define CDX_L_NIN 1
#define CDX_L_NOUT 2
#define CDX_L_SRC 3
#define CDX_L_DST 4
#define CDX_L_WOFF 5
#define CDX_L_BOFF 6
#define CDX_L_FLAGS 7
#define CDX_L_DST2 8
// ---- flags ----
#define CDX_F_GN_MISH 1
#define CDX_F_ADD_EMB 2
#define CDX_F_ADD_RES 4
#define CDX_F_ACCUM 8
#define CDX_F_DST_PRED 16
#define CDX_F_POST_MISH 32
#define CDX_F_RAW_COPY 64
#define CDX_F_SCALE 128
#define CDX_F_KEEP_DST 256
#define CDX_F_FILM 512
// ---- activation ids / normalisation modes ----
#define CDX_ACT_NONE 0
#define CDX_ACT_MISH 1
#define CDX_ACT_GELU_ERF 2
#define CDX_ACT_LEAKY 3
#define CDX_ACT_SILU 4
#define CDX_ACT_RELU 5
#define CDX_ACT_GELU_TANH 6
#define CDX_ACT_MISH_GRAD 7   /* d mish(x) / dx (elementwise map only: classifier-guidance backward) */
#define CDX_ACT_TANH 8        /* critic / inverse-dynamics heads */
#define CDX_NORM_NONE 0
#define CDX_NORM_SLOT_GROUP 1
#define CDX_NORM_COLUMN 2

#define CDX_MODE_16X16 0
#define CDX_MODE_4X4 1
#define CDX_ITEM_WORDS 8
#define CDX_I_WOFF 0
#define CDX_I_PART 1
#define CDX_I_NQ 2
#define CDX_I_ONB 3
#define CDX_I_TAP 4
#define CDX_I_CC 5

#define CDX_HALO 0
#define CDX_N_WAVES 8
#define CDX_GN_EPS 1e-5f
