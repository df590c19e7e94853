// cdx_ops2.h -- word layout of the v2 U-Net program (csrc/cdx_unet2.hip): op descriptors of CDX2_OP_WORDS int32 and
// work-item records of CDX2_ITEM_WORDS int32, both read with scalar loads.
// MUST mirror cleandiffuser_amd/engine/program2.py (tests/test_abi_contract.py parses this file and compares).
#pragma once

#define CDX2_OP_WORDS 48   /* 36 used; 192 B = three 64-B scalar-cache lines */
#define CDX2_ITEM_WORDS 8
#define CDX2_NW2 4        /* waves per workgroup, one per SIMD */
#define CDX2_RING2 16     /* 1-KiB weight records in flight per wave */
#define CDX2_GROUPS2 8    /* GroupNorm groups: 32 lanes per group in the epilogue */
#define CDX2_MAX_NK2 4    /* float4 items one lane may own in the epilogue */

#define CDX2_W2_KIND 0
#define CDX2_W2_FLAGS 1
#define CDX2_W2_COUT 2
#define CDX2_W2_LOUT 3
#define CDX2_W2_LIN 4
#define CDX2_W2_CSTRIDE 5
#define CDX2_W2_TRANSPOSED 6
#define CDX2_W2_MODE 7
#define CDX2_W2_NT 8
#define CDX2_W2_NITEMS 9
#define CDX2_W2_ITEMS 10
#define CDX2_W2_NSEG 11
#define CDX2_W2_SEG0 12
#define CDX2_SEG2_WORDS 5
#define CDX2_S2_SRC 0
#define CDX2_S2_STRIDE 1
#define CDX2_S2_CCN 2
#define CDX2_S2_TAPS 3
#define CDX2_S2_PAD 4
#define CDX2_W2_DST 22
#define CDX2_W2_DST_STRIDE 23
#define CDX2_W2_SSTRIDE 24
#define CDX2_W2_KSPLIT 25
#define CDX2_W2_BOFF 26
#define CDX2_W2_GAMMA 27
#define CDX2_W2_BETA 28
#define CDX2_W2_EMB 29
#define CDX2_W2_RES 30
#define CDX2_W2_RES_STRIDE 31
#define CDX2_W2_CG4_SHIFT 32
#define CDX2_W2_INV_CNT 33
#define CDX2_W2_NK 34
#define CDX2_W2_COUTP 35

#define CDX2_I2_WOFF 0
#define CDX2_I2_NQ 1
#define CDX2_I2_SEG 2
#define CDX2_I2_TAP 3
#define CDX2_I2_CC 4
#define CDX2_I2_PART 5
#define CDX2_I2_COL0 6
#define CDX2_I2_SPARE 7

#define CDX2_F2_GN 1
#define CDX2_F2_EMB 2
#define CDX2_F2_RES 4
#define CDX2_F2_PRED 8
