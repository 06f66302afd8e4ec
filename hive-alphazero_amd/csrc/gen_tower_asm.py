#!/usr/bin/env python3
"""gen_tower_asm.py -- writes the gfx950 assembly of `hive_tower72_{bf16,f16}`: the residual tower of the leaf evaluator
(alpha_net.py:36-54,87-99 of the reference: y = relu(conv2(relu(conv1(x) + b1)) + b2 + x), nblocks times) with a 72-tile
wave -- what hipcc cannot allocate (288 accumulator registers across the AGPR / VGPR halves, profiles/r03_net_tower.md).

Shape (one workgroup = 4 waves = one wave per SIMD = TWO boards):
  * both boards (2 x 144 pixels x 256 channels, 16-bit) live in LDS for the whole tower, pixel stride 544 B, plus a zero
    pixel that every off-board tap reads (157,216 B of the CU's 160 KB);
  * wave w owns output channels 64 w .. 64 w + 63 (4 M tiles) of BOTH boards' 288 pixels (18 N tiles): 72 accumulator
    tiles = a[0:255] + v[..] -- every weight fragment fetched from L2 feeds two boards (half the weight stream per MFMA
    of the one-board kernels, 0.31 instead of 0.36 KiB of operands per MFMA);
  * weights (fragment-major, hive_nn.h) go L2 -> VGPR three k-steps ahead through a ring of four 4-fragment buffers; the
    pixel fragments come from LDS ten reads ahead through a ring of twelve; the 8 k-steps of a tap are unrolled, every LDS
    offset is an immediate, the per-tap pixel offsets (off-board -> zero pixel) come from a table in .rodata;
  * arithmetic and rounding points are exactly those of resblock_kernel / tower_kernel (hive_nn.hip): fp32 accumulation
    in k order, (acc + bias) [+ skip], max(0, .), one rounding to 16 bits -- the results are bit-identical.

The generator keeps a model of the two in-order memory queues (vmcnt / lgkmcnt) and derives every s_waitcnt from it.

One convolution instead of a tower (kernarg flags, see S_FLAGS below): bit 0 = conv + bias [+ ReLU] -> y (hive_nn_conv72), with
bit 2 a residual added in the epilogue (hive_nn_conv72_add), with bit 3 the per-channel sums of the output for a BatchNorm
behind it (hive_nn_conv72_stats): the training step's forward and data-gradient convolutions.

usage: gen_tower_asm.py out.s [debug]     debug builds: 1 = staging identity (LDS image back out), 3 = conv1 + epilogue 1,
                                          4 = lane constants, 5 = s_memtime stamps per block (tools/dev/t72_*.py)
"""
import sys

PS = 544                      # LDS pixel stride in bytes (34 sixteen-byte slots: the 16 lanes of a ds_read_b128 phase hit 16 slots)
NT, MT = 18, 4                # pixel tiles (both boards) x channel tiles per wave
ZOFF = 288 * PS               # the zero pixel
MAILBOX = 289 * PS            # one LDS word behind the image: wave 0's verdict of a hand-over poll
LDS_BYTES = 289 * PS + 64
D = 10                        # B-fragment LDS reads in flight
BRING = 12                    # ring of B-fragment buffers (it lives in the epilogues' temporaries: the phases never overlap)
ARING, AD = 4, 3              # ring of weight buffers; k-steps the weights are fetched ahead
ROW_BYTES = 144 * 256 * 2     # one board in global memory
KSTEP_BYTES = 16 * 1024       # one k-step of packed weights (16 fragments of 1 KiB)
NTP = 20                      # boffN is fetched as five 4-register loads: 18 offsets + 2 pad
TAP_BYTES = (NTP // 4) * 1024 # one tap of the offset table: [group of 4 tiles][lane][4] uint32

# ---- register map -------------------------------------------------------------------------------------------------
V_TID = 0
V_WLANE = 1                   # wave * 4096 + lane * 16: this lane's byte offset inside a k-step's fragments
V_TAB = 2                     # lane * 16
V_BOFF = 4                    # 18 (register tuples must start on even registers): LDS byte offset of this lane's fragment row of pixel tile nt under the current tap
V_BOFFN = V_BOFF + NT         # 20: ... under the next tap
V_A = V_BOFFN + NTP           # 64: weight ring [ARING][MT][4]
V_ACC = V_A + ARING * MT * 4  # 32: accumulator tiles 64..71
V_BIAS = V_ACC + 32           # 16: bias[mt][4]
V_T = V_BIAS + 16             # 30: epilogue temporaries (3 sets of 10)
V_SK = V_T + 30               # 32: skip operands in flight (4 batches x 4 tiles x 2)
V_B = V_T                     # 40: pixel-fragment ring [BRING][4] -- the same registers as V_T / V_SK, used in the tap loops only
assert BRING * 4 <= 30 + 32 and (8 * NT) % BRING == 0 and D < BRING    # (a tap body must end in the ring phase it started in)
V_GO = V_SK + 32              # 4: global pixel offsets of the batches in flight
V_LDSW = V_GO + 4             # 3: epilogue LDS write bases
V_GOFF = V_LDSW + 3           # 1: lr * 512 + lg * 8 + wave * 128
V_BIASOFF = V_GOFF + 1        # 1: wave * 256 + lg * 16
V_FLOOR = V_BIASOFF + 1       # 1: the epilogues' lower clamp: 0.0 (ReLU) or -inf (none)
V_TMP = V_FLOOR + 1           # 4 scratch
V_END = V_TMP + 4
assert V_END <= 256, V_END
NEXT_VGPR = (V_END + 7) // 8 * 8
V_STAGE = V_A                 # 144 registers of staging data (prologue only; everything above V_A is dead then)
assert V_STAGE + 144 <= V_GO

S_KARG = 0
S_WG = 2
S_X, S_W, S_BIAS, S_Y, S_IDX, S_CNT = 4, 6, 8, 10, 12, 14
S_N, S_NBLK = 16, 17
S_SKIP0, S_SKIP1 = 20, 22     # skip operand rows (X for the first block, then Y)
S_Y0, S_Y1 = 24, 26
S_WP = 28                     # weight prefetch pointer (k-step current + AD)
S_WLEFT = 30                  # advances of S_WP still allowed
S_T0 = 31
S_BP = 32                     # bias of the current block's conv1
S_TAP = 34
S_BLK = 35
S_TAB = 36                    # table + 2048
S_TP = 38                     # table pointer of the tap whose offsets are fetched next (+2048)
S_WAVE = 40
S_R0, S_R1 = 41, 42
S_T1, S_T2, S_T3 = 43, 44, 45
S_SRC = 46                    # pair: staging source
S_PLAN = 18                   # pair: the launch plan (or null): per workgroup 8 ints + the hand-over flags behind them
S_D = 56                      # 8: this workgroup's plan entry: head pair, head end block, first full pair, end of the full
                              #    pairs, tail pair, tail first block, -, -
S_PHASE, S_PAIR, S_B0, S_B1, S_MODE, S_POLL = 64, 65, 66, 67, 68, 69
S_FLAG = 70                   # pair: address of the current pair's hand-over flag
S_FLAGS = 76                  # kernarg flags: bit 0 = ONE convolution per block (conv + bias [+ ReLU] -> y), bit 1 = no ReLU,
                              # bit 2 (with bit 0) = ... + residual: the convolution runs as a block's SECOND one, whose epilogue adds a
                              # skip operand -- here the rows of the kernarg `residual` instead of the block's input
                              # bit 3 (with bit 0, without bit 2) = ... and the per-channel sum / sum of squares of the workgroup's
                              # output (as stored: after the rounding) -> aux[workgroup][2][256] floats: the statistics pass of a
                              # training-mode BatchNorm behind the convolution
S_RES = 78                    # pair: kernarg aux: the residual (flags bit 2) or the statistics rows (bit 3)
S_STG0, S_STG1 = 72, 74       # pairs: where the current segment's boards are staged from (X rows, or Y rows for a tail)
PLAN_STRIDE = 32              # bytes per workgroup in the plan
MAX_WG = 256                  # workgroups a plan covers; the flags start at plan + MAX_WG * PLAN_STRIDE
BLOCK_W_BYTES = 2 * 72 * KSTEP_BYTES     # packed weights of one residual block
S_HAS1 = 48                   # pair: exec mask for the second board's global stores (0 when the pair's second entry repeats the first)
NEXT_SGPR = 80
assert S_RES + 2 <= NEXT_SGPR


def acc_reg(t):
    """Register range of accumulator tile t = nt * 4 + mt."""
    return "a[%d:%d]" % (4 * t, 4 * t + 3) if t < 64 else "v[%d:%d]" % (V_ACC + 4 * (t - 64), V_ACC + 4 * (t - 64) + 3)


def vr(base, n=1):
    return "v%d" % base if n == 1 else "v[%d:%d]" % (base, base + n - 1)


def sr(base, n=1):
    return "s%d" % base if n == 1 else "s[%d:%d]" % (base, base + n - 1)


class Asm:
    def __init__(self):
        self.lines = []
        self.vm_q, self.lg_q = [], []          # outstanding tags, oldest first

    def e(self, s):
        self.lines.append("\t" + s)

    def label(self, s):
        self.lines.append(s + ":")

    def comment(self, s):
        self.lines.append("\t; " + s)

    # ---- queue model
    def vm(self, text, tag):
        self.e(text)
        self.vm_q.append(tag)

    def lg(self, text, tag):
        self.e(text)
        self.lg_q.append(tag)

    def wait(self, vm_tag=None, lg_tag=None):
        """Wait until the operations tagged vm_tag / lg_tag (and everything older) have completed."""
        parts = []
        if vm_tag is not None and vm_tag in self.vm_q:
            i = self.vm_q.index(vm_tag)
            n = len(self.vm_q) - 1 - i
            assert n <= 63, n
            parts.append("vmcnt(%d)" % n)
            self.vm_q = self.vm_q[i + 1:]
        if lg_tag is not None and lg_tag in self.lg_q:
            i = self.lg_q.index(lg_tag)
            n = len(self.lg_q) - 1 - i
            assert n <= 15, n
            parts.append("lgkmcnt(%d)" % n)
            self.lg_q = self.lg_q[i + 1:]
        if parts:
            self.e("s_waitcnt " + " ".join(parts))

    def drain(self, vm=True, lg=True):
        parts = []
        if vm:
            parts.append("vmcnt(0)")
            self.vm_q = []
        if lg:
            parts.append("lgkmcnt(0)")
            self.lg_q = []
        self.e("s_waitcnt " + " ".join(parts))


def tap_table():
    """tab[tap][group][lane][4]: LDS byte offsets of lane's B-fragment rows of pixel tiles 4 group .. 4 group + 3 (pixel
    nt * 16 + lane % 16 shifted by the tap, k-group lane / 16); tiles 18, 19 are padding."""
    out = []
    for tap in range(9):
        dy, dx = tap // 3 - 1, tap % 3 - 1
        for grp in range(NTP // 4):
            for lane in range(64):
                for j in range(4):
                    nt = grp * 4 + j
                    if nt >= NT:
                        out.append(ZOFF)
                        continue
                    slot, pt = divmod(nt, 9)
                    lr, lg = lane & 15, lane >> 4
                    pixel = pt * 16 + lr
                    y0, x0 = divmod(pixel, 12)
                    sy, sx = y0 + dy, x0 + dx
                    inb = 0 <= sy < 12 and 0 <= sx < 12
                    out.append(((slot * 144 + sy * 12 + sx) * PS if inb else ZOFF) + lg * 16)
    return out


def gen_kernel(name, dt, debug=0):
    """dt: 'bf16' or 'f16'."""
    A = Asm()
    mfma = "v_mfma_f32_16x16x32_" + dt
    cvt = "v_cvt_pk_bf16_f32" if dt == "bf16" else "v_cvt_pk_f16_f32"
    e, c = A.e, A.comment

    # =============================================================== prologue
    c("kernarg: X W bias Y idx count | batch nblocks | plan | flags | residual")
    e("s_load_dwordx8 %s, %s, 0x0" % (sr(S_X, 8), sr(S_KARG, 2)))
    e("s_load_dwordx4 %s, %s, 0x20" % (sr(S_IDX, 4), sr(S_KARG, 2)))
    e("s_load_dwordx4 %s, %s, 0x30" % (sr(S_N, 4), sr(S_KARG, 2)))
    e("s_load_dword %s, %s, 0x40" % (sr(S_FLAGS), sr(S_KARG, 2)))
    e("s_load_dwordx2 %s, %s, 0x48" % (sr(S_RES, 2), sr(S_KARG, 2)))
    e("v_lshrrev_b32_e32 %s, 6, %s" % (vr(V_TMP), vr(V_TID)))
    e("s_nop 1")                                 # (a VALU-written register is not yet visible to v_readfirstlane)
    e("v_readfirstlane_b32 %s, %s" % (sr(S_WAVE), vr(V_TMP)))
    e("s_nop 4")
    e("s_waitcnt lgkmcnt(0)")
    c("n = count ? *count : batch")
    e("s_cmp_eq_u64 %s, 0" % sr(S_CNT, 2))
    e("s_cbranch_scc1 .L%s_havecount" % name)
    e("s_load_dword %s, %s, 0x0" % (sr(S_N), sr(S_CNT, 2)))
    e("s_waitcnt lgkmcnt(0)")
    A.label(".L%s_havecount" % name)
    c("this workgroup's work: its plan entry, or (no plan) the one pair of boards 2 wg, 2 wg + 1 through every block")
    e("s_mov_b32 %s, -1" % sr(S_D + 0))
    e("s_mov_b32 %s, 0" % sr(S_D + 1))
    e("s_mov_b32 %s, %s" % (sr(S_D + 2), sr(S_WG)))
    e("s_add_u32 %s, %s, 1" % (sr(S_D + 3), sr(S_WG)))
    e("s_mov_b32 %s, -1" % sr(S_D + 4))
    e("s_mov_b32 %s, 0" % sr(S_D + 5))
    e("s_lshl_b32 %s, %s, 1" % (sr(S_T1), sr(S_WG)))
    e("s_cmp_ge_u32 %s, %s" % (sr(S_T1), sr(S_N)))
    e("s_cselect_b32 %s, %s, %s" % (sr(S_D + 3), sr(S_D + 2), sr(S_D + 3)))      # (nothing to do: an empty range of full pairs)
    e("s_cmp_eq_u64 %s, 0" % sr(S_PLAN, 2))
    e("s_cbranch_scc1 .L%s_planned" % name)
    e("s_lshl_b32 %s, %s, %d" % (sr(S_T1), sr(S_WG), PLAN_STRIDE.bit_length() - 1))
    e("s_load_dwordx8 %s, %s, %s" % (sr(S_D, 8), sr(S_PLAN, 2), sr(S_T1)))
    e("s_waitcnt lgkmcnt(0)")
    A.label(".L%s_planned" % name)
    e("s_mov_b32 %s, 0" % sr(S_PHASE))

    c("lane constants")
    LANE, LR, LG = V_TMP + 1, V_TMP + 2, V_TMP + 3
    e("v_and_b32_e32 %s, 63, %s" % (vr(LANE), vr(V_TID)))
    e("v_and_b32_e32 %s, 15, %s" % (vr(LR), vr(V_TID)))
    e("v_bfe_u32 %s, %s, 4, 2" % (vr(LG), vr(V_TID)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_TAB), vr(LANE)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_WLANE), vr(LANE)))
    e("s_lshl_b32 %s, %s, 12" % (sr(S_T1), sr(S_WAVE)))
    e("v_add_u32_e32 %s, %s, %s" % (vr(V_WLANE), sr(S_T1), vr(V_WLANE)))
    e("s_lshl_b32 %s, %s, 7" % (sr(S_T1), sr(S_WAVE)))                    # wave * 128
    e("v_lshlrev_b32_e32 %s, 3, %s" % (vr(V_TMP), vr(LG)))               # lg * 8
    e("v_add_u32_e32 %s, %s, %s" % (vr(V_TMP), sr(S_T1), vr(V_TMP)))      # wave*128 + lg*8
    e("v_mul_u32_u24_e32 %s, 0x%x, %s" % (vr(V_LDSW), PS, vr(LR)))
    e("v_add_u32_e32 %s, %s, %s" % (vr(V_LDSW), vr(V_LDSW), vr(V_TMP)))
    e("v_add_u32_e32 %s, 0x%x, %s" % (vr(V_LDSW + 1), 7 * 16 * PS, vr(V_LDSW)))
    e("v_add_u32_e32 %s, 0x%x, %s" % (vr(V_LDSW + 2), 14 * 16 * PS, vr(V_LDSW)))
    e("v_lshlrev_b32_e32 %s, 9, %s" % (vr(V_GOFF), vr(LR)))
    e("v_add_u32_e32 %s, %s, %s" % (vr(V_GOFF), vr(V_GOFF), vr(V_TMP)))
    e("s_lshl_b32 %s, %s, 8" % (sr(S_T1), sr(S_WAVE)))                    # wave * 256
    e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_BIASOFF), vr(LG)))
    e("v_add_u32_e32 %s, %s, %s" % (vr(V_BIASOFF), sr(S_T1), vr(V_BIASOFF)))

    e("s_bitcmp1_b32 %s, 1" % sr(S_FLAGS))
    e("s_cselect_b32 %s, 0xff800000, 0" % sr(S_T1))
    e("v_mov_b32_e32 %s, %s" % (vr(V_FLOOR), sr(S_T1)))
    c("table of tap offsets (.rodata of this code object), biased by 2048 so that 18 rows fit the 13-bit offsets")
    e("s_getpc_b64 %s" % sr(S_TAB, 2))
    e("s_add_u32 %s, %s, hive_tap_table@rel32@lo+2052" % (sr(S_TAB), sr(S_TAB)))
    e("s_addc_u32 %s, %s, hive_tap_table@rel32@hi+2060" % (sr(S_TAB + 1), sr(S_TAB + 1)))

    # =============================================================== the workgroup's segments, one after the other
    A.label(".L%s_next" % name)
    c("phase 0: the head of a pair that another workgroup finishes (first, so that its successor never waits long)")
    e("s_cmp_lg_u32 %s, 0" % sr(S_PHASE))
    e("s_cbranch_scc1 .L%s_ph1" % name)
    e("s_mov_b32 %s, 1" % sr(S_PHASE))
    e("s_cmp_lt_i32 %s, 0" % sr(S_D + 0))
    e("s_cbranch_scc1 .L%s_ph1" % name)
    e("s_mov_b32 %s, %s" % (sr(S_PAIR), sr(S_D + 0)))
    e("s_mov_b32 %s, 0" % sr(S_B0))
    e("s_mov_b32 %s, %s" % (sr(S_B1), sr(S_D + 1)))
    e("s_mov_b32 %s, 2" % sr(S_MODE))
    e("s_branch .L%s_go" % name)
    A.label(".L%s_ph1" % name)
    c("phase 1: whole pairs")
    e("s_cmp_lg_u32 %s, 1" % sr(S_PHASE))
    e("s_cbranch_scc1 .L%s_ph2" % name)
    e("s_cmp_ge_i32 %s, %s" % (sr(S_D + 2), sr(S_D + 3)))
    e("s_cbranch_scc1 .L%s_ph1done" % name)
    e("s_mov_b32 %s, %s" % (sr(S_PAIR), sr(S_D + 2)))
    e("s_add_u32 %s, %s, 1" % (sr(S_D + 2), sr(S_D + 2)))
    e("s_mov_b32 %s, 0" % sr(S_B0))
    e("s_mov_b32 %s, %s" % (sr(S_B1), sr(S_NBLK)))
    e("s_mov_b32 %s, 0" % sr(S_MODE))
    e("s_branch .L%s_go" % name)
    A.label(".L%s_ph1done" % name)
    e("s_mov_b32 %s, 2" % sr(S_PHASE))
    A.label(".L%s_ph2" % name)
    c("phase 2: the tail of a pair whose head another workgroup computed first thing")
    e("s_cmp_lg_u32 %s, 2" % sr(S_PHASE))
    e("s_cbranch_scc1 .L%s_end" % name)
    e("s_mov_b32 %s, 3" % sr(S_PHASE))
    e("s_cmp_lt_i32 %s, 0" % sr(S_D + 4))
    e("s_cbranch_scc1 .L%s_end" % name)
    e("s_mov_b32 %s, %s" % (sr(S_PAIR), sr(S_D + 4)))
    e("s_mov_b32 %s, %s" % (sr(S_B0), sr(S_D + 5)))
    e("s_mov_b32 %s, %s" % (sr(S_B1), sr(S_NBLK)))
    e("s_mov_b32 %s, 1" % sr(S_MODE))
    A.label(".L%s_go" % name)
    c("rows of the pair; an odd tail repeats the first board (its copy computes along and stores nothing)")
    e("s_lshl_b32 %s, %s, 1" % (sr(S_R0), sr(S_PAIR)))
    e("s_add_u32 %s, %s, 1" % (sr(S_R1), sr(S_R0)))
    e("s_cmp_ge_u32 %s, %s" % (sr(S_R1), sr(S_N)))
    e("s_cselect_b32 %s, %s, %s" % (sr(S_R1), sr(S_R0), sr(S_R1)))
    e("s_cselect_b64 %s, 0, -1" % sr(S_HAS1, 2))      # exec mask of the second board's stores: nothing if it is a repeat
    e("s_cmp_eq_u64 %s, 0" % sr(S_IDX, 2))
    e("s_cbranch_scc1 .L%s_rows" % name)
    if debug == 5:
        e("s_branch .L%s_rows" % name)
    e("s_lshl_b32 %s, %s, 2" % (sr(S_T1), sr(S_R0)))
    e("s_lshl_b32 %s, %s, 2" % (sr(S_T2), sr(S_R1)))
    e("s_load_dword %s, %s, %s" % (sr(S_R0), sr(S_IDX, 2), sr(S_T1)))
    e("s_load_dword %s, %s, %s" % (sr(S_R1), sr(S_IDX, 2), sr(S_T2)))
    e("s_waitcnt lgkmcnt(0)")
    A.label(".L%s_rows" % name)
    c("row byte offsets (64 bit) -> X and Y rows of the two boards")
    for r, skip, yb in ((S_R0, S_SKIP0, S_Y0), (S_R1, S_SKIP1, S_Y1)):
        e("s_mul_hi_u32 %s, %s, 0x%x" % (sr(S_T2), sr(r), ROW_BYTES))
        e("s_mul_i32 %s, %s, 0x%x" % (sr(S_T1), sr(r), ROW_BYTES))
        e("s_add_u32 %s, %s, %s" % (sr(skip), sr(S_X), sr(S_T1)))
        e("s_addc_u32 %s, %s, %s" % (sr(skip + 1), sr(S_X + 1), sr(S_T2)))
        e("s_add_u32 %s, %s, %s" % (sr(yb), sr(S_Y), sr(S_T1)))
        e("s_addc_u32 %s, %s, %s" % (sr(yb + 1), sr(S_Y + 1), sr(S_T2)))
    c("the previous segment's image -> Y reads of LDS are done in every wave before the image is overwritten")
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    c("a tail starts from block b0 - 1's output (Y), written by another workgroup: wait for its flag, then acquire")
    e("s_cmp_eq_u32 %s, 0" % sr(S_B0))
    e("s_cbranch_scc1 .L%s_fromx" % name)
    e("s_lshl_b32 %s, %s, 2" % (sr(S_T1), sr(S_PAIR)))
    e("s_add_u32 %s, %s, 0x%x" % (sr(S_FLAG), sr(S_PLAN), MAX_WG * PLAN_STRIDE))
    e("s_addc_u32 %s, %s, 0" % (sr(S_FLAG + 1), sr(S_PLAN + 1)))
    e("s_add_u32 %s, %s, %s" % (sr(S_FLAG), sr(S_FLAG), sr(S_T1)))
    e("s_addc_u32 %s, %s, 0" % (sr(S_FLAG + 1), sr(S_FLAG + 1)))
    c("wave 0 polls (bounded); its verdict reaches the other waves through an LDS word, so that all four take the same path")
    e("s_cmp_lg_u32 %s, 0" % sr(S_WAVE))
    e("s_cbranch_scc1 .L%s_pollwait" % name)
    e("s_mov_b32 %s, 0" % sr(S_POLL))
    e("v_mov_b32_e32 %s, 0" % vr(V_TMP))
    A.label(".L%s_poll" % name)
    e("global_load_dword %s, %s, %s sc1" % (vr(V_TMP + 1), vr(V_TMP), sr(S_FLAG, 2)))
    e("s_waitcnt vmcnt(0)")
    e("v_readfirstlane_b32 %s, %s" % (sr(S_T1), vr(V_TMP + 1)))
    e("s_nop 4")
    e("s_cmp_ge_i32 %s, %s" % (sr(S_T1), sr(S_B0)))
    e("s_cselect_b32 %s, %s, 0" % (sr(S_T2), sr(S_B0)))            # the verdict so far: b0 (flag seen) or 0
    e("s_cbranch_scc1 .L%s_polled" % name)
    e("s_sleep 32")
    e("s_add_u32 %s, %s, 1" % (sr(S_POLL), sr(S_POLL)))
    e("s_cmp_lt_u32 %s, 0x%x" % (sr(S_POLL), 1 << 19))
    e("s_cbranch_scc1 .L%s_poll" % name)
    c("(the head never came -- its workgroup is not running: verdict 0 = this workgroup computes the whole pair itself; both write the same bytes)")
    A.label(".L%s_polled" % name)
    e("v_mov_b32_e32 %s, %s" % (vr(V_TMP + 1), sr(S_T2)))
    e("v_mov_b32_e32 %s, 0x%x" % (vr(V_TMP), MAILBOX))
    e("ds_write_b32 %s, %s" % (vr(V_TMP), vr(V_TMP + 1)))
    e("s_waitcnt lgkmcnt(0)")
    A.label(".L%s_pollwait" % name)
    e("s_barrier")
    e("v_mov_b32_e32 %s, 0x%x" % (vr(V_TMP), MAILBOX))
    e("ds_read_b32 %s, %s" % (vr(V_TMP + 1), vr(V_TMP)))
    e("s_waitcnt lgkmcnt(0)")
    e("v_readfirstlane_b32 %s, %s" % (sr(S_B0), vr(V_TMP + 1)))
    e("s_nop 4")
    e("s_cmp_eq_u32 %s, 0" % sr(S_B0))
    e("s_cbranch_scc1 .L%s_fromx" % name)
    A.label(".L%s_acquire" % name)
    e("buffer_inv sc1")
    e("s_waitcnt vmcnt(0)")
    e("s_mov_b64 %s, %s" % (sr(S_SKIP0, 2), sr(S_Y0, 2)))
    e("s_mov_b64 %s, %s" % (sr(S_SKIP1, 2), sr(S_Y1, 2)))
    A.label(".L%s_fromx" % name)
    e("s_mov_b64 %s, %s" % (sr(S_STG0, 2), sr(S_SKIP0, 2)))
    e("s_mov_b64 %s, %s" % (sr(S_STG1, 2), sr(S_SKIP1, 2)))
    c("flags bit 2: the skip operand of the (one) convolution = the same rows of `residual`")
    e("s_bitcmp1_b32 %s, 2" % sr(S_FLAGS))
    e("s_cbranch_scc0 .L%s_nores" % name)
    for skip, stg in ((S_SKIP0, S_STG0), (S_SKIP1, S_STG1)):
        e("s_sub_u32 %s, %s, %s" % (sr(S_T1), sr(stg), sr(S_X)))
        e("s_subb_u32 %s, %s, %s" % (sr(S_T2), sr(stg + 1), sr(S_X + 1)))
        e("s_add_u32 %s, %s, %s" % (sr(skip), sr(S_RES), sr(S_T1)))
        e("s_addc_u32 %s, %s, %s" % (sr(skip + 1), sr(S_RES + 1), sr(S_T2)))
    A.label(".L%s_nores" % name)
    c("this segment's blocks: weights, biases, counters")
    e("s_mul_i32 %s, %s, 0x%x" % (sr(S_T1), sr(S_B0), BLOCK_W_BYTES))
    e("s_mul_hi_u32 %s, %s, 0x%x" % (sr(S_T2), sr(S_B0), BLOCK_W_BYTES))
    e("s_add_u32 %s, %s, %s" % (sr(S_WP), sr(S_W), sr(S_T1)))
    e("s_addc_u32 %s, %s, %s" % (sr(S_WP + 1), sr(S_W + 1), sr(S_T2)))
    e("s_lshl_b32 %s, %s, 11" % (sr(S_T1), sr(S_B0)))
    e("s_add_u32 %s, %s, %s" % (sr(S_BP), sr(S_BIAS), sr(S_T1)))
    e("s_addc_u32 %s, %s, 0" % (sr(S_BP + 1), sr(S_BIAS + 1)))
    e("s_bitcmp1_b32 %s, 2" % sr(S_FLAGS))               # (the second convolution's epilogue reads its bias 1 KiB further on)
    e("s_cselect_b32 %s, 0x400, 0" % sr(S_T1))
    e("s_sub_u32 %s, %s, %s" % (sr(S_BP), sr(S_BP), sr(S_T1)))
    e("s_subb_u32 %s, %s, 0" % (sr(S_BP + 1), sr(S_BP + 1)))
    e("s_sub_u32 %s, %s, %s" % (sr(S_BLK), sr(S_B1), sr(S_B0)))
    e("s_sub_u32 %s, %s, %s" % (sr(S_WLEFT), sr(S_NBLK), sr(S_B0)))
    e("s_mul_i32 %s, %s, 144" % (sr(S_WLEFT), sr(S_WLEFT)))
    e("s_sub_u32 %s, %s, %d" % (sr(S_WLEFT), sr(S_WLEFT), AD + 1))
    e("s_bitcmp1_b32 %s, 0" % sr(S_FLAGS))               # ONE convolution in all (b0 = 0, one block): its 72 k-steps
    e("s_cselect_b32 %s, %d, %s" % (sr(S_WLEFT), 72 - AD - 1, sr(S_WLEFT)))

    c("stage both boards: 36 sixteen-byte loads per thread, all in flight, then 36 LDS writes")
    ST_ADDR, ST_G = V_BOFF, V_BOFF + 1          # (V_BOFF.. are dead until the first convolution)
    e("v_lshrrev_b32_e32 %s, 5, %s" % (vr(ST_ADDR), vr(V_TID)))
    e("v_mul_u32_u24_e32 %s, 0x%x, %s" % (vr(ST_ADDR), PS, vr(ST_ADDR)))
    e("v_and_b32_e32 %s, 31, %s" % (vr(V_TMP), vr(V_TID)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_TMP), vr(V_TMP)))
    e("v_add_u32_e32 %s, %s, %s" % (vr(ST_ADDR), vr(ST_ADDR), vr(V_TMP)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(ST_G), vr(V_TID)))
    for s_, base in ((0, S_STG0), (1, S_STG1)):
        e("s_mov_b64 %s, %s" % (sr(S_SRC, 2), sr(base, 2)))
        for j in range(18):
            k = s_ * 18 + j
            A.vm("global_load_dwordx4 %s, %s, %s" % (vr(V_STAGE + 4 * k, 4), vr(ST_G), sr(S_SRC, 2)), ("stage", k))
            if j < 17:
                e("s_add_u32 %s, %s, 0x1000" % (sr(S_SRC), sr(S_SRC)))
                e("s_addc_u32 %s, %s, 0" % (sr(S_SRC + 1), sr(S_SRC + 1)))
    c("zero pixel (threads 0..33; the others repeat the last slot) and the accumulators while the loads fly")
    e("v_min_u32_e32 %s, 33, %s" % (vr(V_TMP), vr(V_TID)))
    e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_TMP), vr(V_TMP)))
    e("v_add_u32_e32 %s, 0x%x, %s" % (vr(V_TMP), ZOFF, vr(V_TMP)))
    Z = V_TAB + 0  # placeholder (unused)
    ZR = V_BOFF + 4                              # four zero registers (dead range)
    for i in range(4):
        e("v_mov_b32_e32 %s, 0" % vr(ZR + i))
    A.lg("ds_write_b128 %s, %s" % (vr(V_TMP), vr(ZR, 4)), ("zero",))
    for k in range(36):
        s_, j = divmod(k, 18)
        cst = (8 * j + 144 * s_) * PS
        hi, lo = cst // 32768 * 32768, cst % 32768
        addr = vr(ST_ADDR)
        if hi:
            e("v_add_u32_e32 %s, 0x%x, %s" % (vr(V_TMP + 1), hi, vr(ST_ADDR)))
            addr = vr(V_TMP + 1)
        A.wait(vm_tag=("stage", k))
        A.lg("ds_write_b128 %s, %s offset:%d" % (addr, vr(V_STAGE + 4 * k, 4), lo), ("stw", k))
        if len(A.lg_q) > 12:
            A.wait(lg_tag=A.lg_q[-8])

    def dump_lds_and_stop():
        A.drain()
        e("s_barrier")
        ST_ADDR, ST_G = V_TMP + 2, V_TMP + 3         # (recomputed: the prologue's copies lived in the boff registers)
        e("v_lshrrev_b32_e32 %s, 5, %s" % (vr(ST_ADDR), vr(V_TID)))
        e("v_mul_u32_u24_e32 %s, 0x%x, %s" % (vr(ST_ADDR), PS, vr(ST_ADDR)))
        e("v_and_b32_e32 %s, 31, %s" % (vr(V_TMP), vr(V_TID)))
        e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_TMP), vr(V_TMP)))
        e("v_add_u32_e32 %s, %s, %s" % (vr(ST_ADDR), vr(ST_ADDR), vr(V_TMP)))
        e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(ST_G), vr(V_TID)))
        for s_, base in ((0, S_Y0), (1, S_Y1)):
            e("s_mov_b64 %s, %s" % (sr(S_SRC, 2), sr(base, 2)))
            for j in range(18):
                cst = (8 * j + 144 * s_) * PS
                hi, lo = cst // 32768 * 32768, cst % 32768
                addr = vr(ST_ADDR)
                if hi:
                    e("v_add_u32_e32 %s, 0x%x, %s" % (vr(V_TMP + 1), hi, vr(ST_ADDR)))
                    addr = vr(V_TMP + 1)
                e("ds_read_b128 %s, %s offset:%d" % (vr(V_STAGE, 4), addr, lo))
                e("s_waitcnt lgkmcnt(0)")
                e("global_store_dwordx4 %s, %s, %s" % (vr(ST_G), vr(V_STAGE, 4), sr(S_SRC, 2)))
                e("s_nop 1")
                e("s_add_u32 %s, %s, 0x1000" % (sr(S_SRC), sr(S_SRC)))
                e("s_addc_u32 %s, %s, 0" % (sr(S_SRC + 1), sr(S_SRC + 1)))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")
        A.label(".L%s_end" % name)
        e("s_endpgm")
        return A.lines

    if debug == 1:
        c("DEBUG 1: copy the LDS image back out (identity) and stop")
        return dump_lds_and_stop()
    if debug == 4:
        c("DEBUG 4: every thread stores its lane constants (16 dwords at Y + tid * 64) and stops")
        A.drain()
        regs = [V_TID, V_WLANE, V_TAB, V_LDSW, V_LDSW + 1, V_LDSW + 2, V_GOFF, V_BIASOFF]
        e("v_lshlrev_b32_e32 %s, 6, %s" % (vr(V_TMP), vr(V_TID)))
        for i, r in enumerate(regs):
            e("global_store_dword %s, %s, %s offset:%d" % (vr(V_TMP), vr(r), sr(S_Y, 2), 4 * i))
        for i, sreg in enumerate((S_WAVE, S_WG, S_N, S_NBLK, S_R0, S_R1, S_Y0, S_Y1)):
            e("v_mov_b32_e32 %s, %s" % (vr(V_TMP + 1), sr(sreg)))
            e("global_store_dword %s, %s, %s offset:%d" % (vr(V_TMP), vr(V_TMP + 1), sr(S_Y, 2), 32 + 4 * i))
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")
        A.label(".L%s_end" % name)
        e("s_endpgm")
        return A.lines

    c("weights: the segment's k-steps 0 .. AD-1 into ring buffers 0 .. AD-1; S_WP -> k-step AD")
    for j in range(AD):
        for mt in range(MT):
            A.vm("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(V_A + (j * MT + mt) * 4, 4), vr(V_WLANE), sr(S_WP, 2), mt * 1024),
                 ("A0", j, mt))
        e("s_add_u32 %s, %s, 0x%x" % (sr(S_WP), sr(S_WP), KSTEP_BYTES))
        e("s_addc_u32 %s, %s, 0" % (sr(S_WP + 1), sr(S_WP + 1)))
    A.drain()
    e("s_barrier")

    # =============================================================== one convolution
    def conv(tag):
        c("---- convolution %s: 9 taps x 8 k-steps x (18 pixel tiles x 4 channel tiles)" % tag)
        c("tap 0 offsets -> boff; table pointer -> tap 1")
        for grp in range(NTP // 4):
            n = 4 if grp * 4 + 4 <= NT else NT - grp * 4      # (the padded entries of the last group go to the first boffN registers)
            dst = V_BOFF + grp * 4
            if n == 4:
                A.vm("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(dst, 4), vr(V_TAB), sr(S_TAB, 2), grp * 1024 - 2048), ("T0", grp))
            else:
                A.vm("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(dst, 4), vr(V_TAB), sr(S_TAB, 2), grp * 1024 - 2048), ("T0", grp))
        e("s_add_u32 %s, %s, 0x%x" % (sr(S_TP), sr(S_TAB), TAP_BYTES))
        e("s_addc_u32 %s, %s, 0" % (sr(S_TP + 1), sr(S_TAB + 1)))
        e("s_mov_b32 %s, 1" % sr(S_TAP))
        A.drain(vm=True, lg=False)
        for g in range(D):
            A.lg("ds_read_b128 %s, %s%s" % (vr(V_B + (g % BRING) * 4, 4), vr(V_BOFF + g % NT),
                                            " offset:%d" % ((g // NT) * 64) if g // NT else ""), ("B", g))

        def tap_body(first):
            T = Asm()                                # generated in its steady state: the queues as every iteration finds them
            T.lines = A.lines
            T.vm_q = [("A", j, mt) for j in range(AD) for mt in range(MT)]
            T.lg_q = [("B", g) for g in range(D)]
            for f in range(8 * NT):
                kc, nt = divmod(f, NT)
                if nt == 0:
                    T.wait(vm_tag=("A", kc, MT - 1))
                g = f + D
                kcg, ntg = divmod(g, NT)
                # (reads of k-step 8 belong to the next tap's first k-step: boff[ntg] holds the next tap's offset by then)
                T.lg("ds_read_b128 %s, %s%s" % (vr(V_B + (g % BRING) * 4, 4), vr(V_BOFF + ntg),
                                                " offset:%d" % ((kcg % 8) * 64) if kcg % 8 else ""), ("B", g))
                if kcg == 7:
                    T.wait(vm_tag=("T", ntg // 4))
                    T.e("v_mov_b32_e32 %s, %s" % (vr(V_BOFF + ntg), vr(V_BOFFN + ntg)))
                T.wait(lg_tag=("B", f))
                for mt in range(MT):
                    t = nt * MT + mt
                    T.e("%s %s, %s, %s, %s" % (mfma, acc_reg(t), vr(V_A + ((kc % ARING) * MT + mt) * 4, 4),
                                                 vr(V_B + (f % BRING) * 4, 4), "0" if first and kc == 0 else acc_reg(t)))
                    if kc == 0 and mt == 1 and nt < NTP // 4:
                        T.vm("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(V_BOFFN + nt * 4, 4), vr(V_TAB), sr(S_TP, 2),
                                                                          nt * 1024 - 2048), ("T", nt))
                    if nt in (2, 6, 10, 14) and mt == 2:
                        m = (nt - 2) // 4
                        T.vm("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(V_A + (((kc + AD) % ARING) * MT + m) * 4, 4),
                                                                          vr(V_WLANE), sr(S_WP, 2), m * 1024), ("A", kc + AD, m))
                if nt == 16:
                    T.comment("advance the weight pointer by one k-step unless it already points at the last one")
                    T.e("s_cmp_lg_u32 %s, 0" % sr(S_WLEFT))
                    T.e("s_cselect_b32 %s, 0x%x, 0" % (sr(S_T0), KSTEP_BYTES))
                    T.e("s_cselect_b32 %s, 1, 0" % sr(S_T1))
                    T.e("s_add_u32 %s, %s, %s" % (sr(S_WP), sr(S_WP), sr(S_T0)))
                    T.e("s_addc_u32 %s, %s, 0" % (sr(S_WP + 1), sr(S_WP + 1)))
                    T.e("s_sub_u32 %s, %s, %s" % (sr(S_WLEFT), sr(S_WLEFT), sr(S_T1)))
            # steady state reached again?
            assert T.vm_q == [("A", 8 + j, mt) for j in range(AD) for mt in range(MT)], T.vm_q
            assert T.lg_q == [("B", 8 * NT + g) for g in range(D)], T.lg_q
            T.comment("next tap: table pointer (wraps to tap 0 after the last one: those offsets are fetched and never used)")
            T.e("s_add_u32 %s, %s, 1" % (sr(S_TAP), sr(S_TAP)))
            T.e("s_add_u32 %s, %s, 0x%x" % (sr(S_TP), sr(S_TP), TAP_BYTES))
            T.e("s_addc_u32 %s, %s, 0" % (sr(S_TP + 1), sr(S_TP + 1)))
            T.e("s_cmp_eq_u32 %s, 9" % sr(S_TAP))
            T.e("s_cselect_b32 %s, %s, %s" % (sr(S_TP), sr(S_TAB), sr(S_TP)))
            T.e("s_cselect_b32 %s, %s, %s" % (sr(S_TP + 1), sr(S_TAB + 1), sr(S_TP + 1)))

        c("tap 0, peeled: its first k-step starts every accumulator from the constant 0 (no zeroing pass anywhere)")
        tap_body(True)
        A.label(".L%s_%s_tap" % (name, tag))
        tap_body(False)
        e("s_cmp_lt_u32 %s, 10" % sr(S_TAP))
        e("s_cbranch_scc1 .L%s_%s_tap" % (name, tag))
        c("drain the stray fragment reads, let the last MFMAs retire")
        e("s_waitcnt lgkmcnt(0)")
        for _ in range(3):
            e("s_nop 7")
        A.vm_q, A.lg_q = [], []                      # (the A prefetches in flight are drained by the epilogue's vmcnt(0))

    def tile_src(t, tmp):
        """Accumulator tile t -> four consecutive VGPRs holding it (AGPR tiles are read into tmp)."""
        if t < 64:
            for i in range(4):
                e("v_accvgpr_read_b32 %s, a%d" % (vr(tmp + i), 4 * t + i))
            return tmp
        return V_ACC + 4 * (t - 64)

    def bias_loads(extra):
        for mt in range(MT):
            A.vm("global_load_dwordx4 %s, %s, %s offset:%d" % (vr(V_BIAS + 4 * mt, 4), vr(V_BIASOFF), sr(S_BP, 2), mt * 64 + extra),
                 ("bias", mt))

    S_STAMP = 50                                     # pair: s_memtime value; S_SP = 52: pair: where this workgroup's stamps go

    def stamp(k):
        """DEBUG 5: wave 0 stores s_memtime at point k of the block (the queues are empty at every call site)."""
        if debug != 5:
            return
        e("s_memtime %s" % sr(S_STAMP, 2))
        e("s_waitcnt lgkmcnt(0)")
        e("s_cmp_lg_u32 %s, 0" % sr(S_WAVE))
        e("s_cbranch_scc1 .L%s_nostamp%d" % (name, stamp.n))
        pair, adr = V_GO, V_GO + 2                   # (free registers: an even pair + one)
        e("v_mov_b32_e32 %s, %s" % (vr(pair), sr(S_STAMP)))
        e("v_mov_b32_e32 %s, %s" % (vr(pair + 1), sr(S_STAMP + 1)))
        e("v_mov_b32_e32 %s, 0" % vr(adr))
        e("global_store_dwordx2 %s, %s, %s offset:%d" % (vr(adr), vr(pair, 2), sr(52, 2), 8 * k))
        e("s_waitcnt vmcnt(0)")
        A.label(".L%s_nostamp%d" % (name, stamp.n))
        stamp.n += 1
    stamp.n = 0

    if debug == 5:
        c("DEBUG 5: stamps go to rows[] (the row list is ignored by this build): [workgroup][block][16] uint64")
        e("s_mul_i32 %s, %s, %s" % (sr(S_T1), sr(S_WG), sr(S_NBLK)))
        e("s_lshl_b32 %s, %s, 7" % (sr(S_T1), sr(S_T1)))
        e("s_add_u32 52, %s, %s" % (sr(S_IDX), sr(S_T1)) if False else "s_add_u32 s52, %s, %s" % (sr(S_IDX), sr(S_T1)))
        e("s_addc_u32 s53, %s, 0" % sr(S_IDX + 1))
    GROUPS = [V_BOFF + 4 * i for i in range((NT + NTP) // 4)] + [V_T + 4 * i for i in range((30 + 32) // 4)]
    assert len(GROUPS) >= 18
    ST_ADDR, ST_G = V_TMP + 2, V_TMP + 3

    def st_regs():
        e("v_lshrrev_b32_e32 %s, 5, %s" % (vr(ST_ADDR), vr(V_TID)))
        e("v_mul_u32_u24_e32 %s, 0x%x, %s" % (vr(ST_ADDR), PS, vr(ST_ADDR)))
        e("v_and_b32_e32 %s, 31, %s" % (vr(V_TMP), vr(V_TID)))
        e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(V_TMP), vr(V_TMP)))
        e("v_add_u32_e32 %s, %s, %s" % (vr(ST_ADDR), vr(ST_ADDR), vr(V_TMP)))
        e("v_lshlrev_b32_e32 %s, 4, %s" % (vr(ST_G), vr(V_TID)))

    def image_addr(slot, j):
        """(address register, immediate) of this thread's j-th 16-byte piece of board `slot` in the LDS image."""
        cst = (8 * j + 144 * slot) * PS
        hi, lo = cst // 32768 * 32768, cst % 32768
        if hi:
            e("v_add_u32_e32 %s, 0x%x, %s" % (vr(V_TMP + 1), hi, vr(ST_ADDR)))
            return vr(V_TMP + 1), lo
        return vr(ST_ADDR), lo

    def image_to_y():
        c("image -> Y, sixteen bytes per lane, 1 KiB per wave-instruction")
        PIPE = 9
        order = [(slot, j) for slot in (0, 1) for j in range(18)]

        def out_read(i):
            slot, j = order[i]
            adr, lo = image_addr(slot, j)
            A.lg("ds_read_b128 %s, %s offset:%d" % (vr(GROUPS[i % 18], 4), adr, lo), ("or", i))

        for i in range(PIPE):
            out_read(i)
        for i, (slot, j) in enumerate(order):
            if j == 0:
                e("s_mov_b64 %s, %s" % (sr(S_SRC, 2), sr(S_Y1 if slot else S_Y0, 2)))
                if slot:
                    e("s_mov_b64 exec, %s" % sr(S_HAS1, 2))          # (a repeated tail board is stored once)
            A.wait(lg_tag=("or", i))
            A.vm("global_store_dwordx4 %s, %s, %s" % (vr(ST_G), vr(GROUPS[i % 18], 4), sr(S_SRC, 2)), ("st", i))
            if j < 17:
                e("s_add_u32 %s, %s, 0x1000" % (sr(S_SRC), sr(S_SRC)))
                e("s_addc_u32 %s, %s, 0" % (sr(S_SRC + 1), sr(S_SRC + 1)))
            if i + PIPE < len(order):
                if i + PIPE >= 18:
                    A.wait(vm_tag=("st", i + PIPE - 18))             # (the register group's previous store has read its data)
                out_read(i + PIPE)
        e("s_mov_b64 exec, -1")

    A.label(".L%s_block" % name)
    stamp(0)
    c("flags bit 2: one convolution WITH a residual = the second convolution's code on the staged boards and the first weights")
    e("s_bitcmp1_b32 %s, 2" % sr(S_FLAGS))
    e("s_cbranch_scc1 .L%s_twoconv" % name)
    # =============================================================== conv1 + epilogue 1
    conv("c1")
    c("---- epilogue 1: relu(acc + b1) -> 16 bits -> over the boards in LDS")
    stamp(1)
    bias_loads(0)
    e("s_barrier")                                   # every wave has finished reading the block's input from LDS
    if debug == 5:
        A.drain()
        stamp(2)
    A.wait(vm_tag=("bias", MT - 1))

    def epilogue1_tiles(stats):
        """stats: also add every stored value (the 16-bit result, widened again) and its square into this lane's 2 x 16 sums
        V_SK[0:15] (sum) / V_SK[16:31] (squares), indexed [channel tile][row of the lane]; V_SK is idle in this epilogue."""
        k = 0
        for nt in range(NT):
            for mt in range(MT):
                t = nt * MT + mt
                tmp = V_T + 10 * (k % 3)
                k += 1
                src = tile_src(t, tmp)
                e("v_pk_add_f32 %s, %s, %s" % (vr(tmp + 4, 2), vr(src, 2), vr(V_BIAS + 4 * mt, 2)))
                e("v_pk_add_f32 %s, %s, %s" % (vr(tmp + 6, 2), vr(src + 2, 2), vr(V_BIAS + 4 * mt + 2, 2)))
                for i in range(4):
                    e("v_max_f32_e32 %s, %s, %s" % (vr(tmp + 4 + i), vr(V_FLOOR), vr(tmp + 4 + i)))
                e("%s %s, %s, %s" % (cvt, vr(tmp + 8), vr(tmp + 4), vr(tmp + 5)))
                e("%s %s, %s, %s" % (cvt, vr(tmp + 9), vr(tmp + 6), vr(tmp + 7)))
                if len(A.lg_q) >= 3:
                    A.wait(lg_tag=A.lg_q[-3])            # (the temporaries of three tiles ago are free again)
                A.lg("ds_write_b64 %s, %s offset:%d" % (vr(V_LDSW + nt // 7), vr(tmp + 8, 2), (nt % 7) * 16 * PS + mt * 32), ("w1", t))
                if not stats:
                    continue
                if dt == "bf16":
                    e("v_lshlrev_b32_e32 %s, 16, %s" % (vr(tmp + 0), vr(tmp + 8)))
                    e("v_and_b32_e32 %s, 0xffff0000, %s" % (vr(tmp + 1), vr(tmp + 8)))
                    e("v_lshlrev_b32_e32 %s, 16, %s" % (vr(tmp + 2), vr(tmp + 9)))
                    e("v_and_b32_e32 %s, 0xffff0000, %s" % (vr(tmp + 3), vr(tmp + 9)))
                else:
                    e("v_cvt_f32_f16_e32 %s, %s" % (vr(tmp + 0), vr(tmp + 8)))
                    e("v_cvt_f32_f16_sdwa %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (vr(tmp + 1), vr(tmp + 8)))
                    e("v_cvt_f32_f16_e32 %s, %s" % (vr(tmp + 2), vr(tmp + 9)))
                    e("v_cvt_f32_f16_sdwa %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (vr(tmp + 3), vr(tmp + 9)))
                sm, sq = V_SK + 4 * mt, V_SK + 16 + 4 * mt
                for h in (0, 2):
                    if nt == 0:                          # (the first pixel tile starts the sums: no zeroing pass)
                        e("v_pk_mul_f32 %s, %s, %s" % (vr(sq + h, 2), vr(tmp + h, 2), vr(tmp + h, 2)))
                        e("v_mov_b32_e32 %s, %s" % (vr(sm + h), vr(tmp + h)))
                        e("v_mov_b32_e32 %s, %s" % (vr(sm + h + 1), vr(tmp + h + 1)))
                    else:
                        e("v_pk_fma_f32 %s, %s, %s, %s" % (vr(sq + h, 2), vr(tmp + h, 2), vr(tmp + h, 2), vr(sq + h, 2)))
                        e("v_pk_add_f32 %s, %s, %s" % (vr(sm + h, 2), vr(sm + h, 2), vr(tmp + h, 2)))

    c("flags bit 3: the statistics variant of this epilogue")
    e("s_bitcmp1_b32 %s, 3" % sr(S_FLAGS))
    e("s_cbranch_scc1 .L%s_epi1stats" % name)
    epilogue1_tiles(False)
    A.drain()
    e("s_branch .L%s_epi1done" % name)
    A.label(".L%s_epi1stats" % name)
    epilogue1_tiles(True)
    A.drain()
    c("sums over the 16 lanes of a row (= the 16 pixels of a tile row): inclusive prefix sums, lane 15 of every row holds the total")
    for shift in (1, 2, 4, 8):
        for r in range(32):
            e("v_add_f32_dpp %s, %s, %s row_shr:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % (vr(V_SK + r), vr(V_SK + r), vr(V_SK + r), shift))
    c("lanes 15 / 31 / 47 / 63 store: aux[workgroup][0][channel] = sum, [1][channel] = squares; channel = wave * 64 + tile * 16 + 4 * row group + i")
    e("s_lshl_b32 %s, %s, 11" % (sr(S_T1), sr(S_PAIR)))
    e("s_add_u32 %s, %s, %s" % (sr(S_SRC), sr(S_RES), sr(S_T1)))
    e("s_addc_u32 %s, %s, 0" % (sr(S_SRC + 1), sr(S_RES + 1)))
    e("s_mov_b32 exec_lo, 0x80008000")
    e("s_mov_b32 exec_hi, 0x80008000")
    for mt in range(MT):
        A.vm("global_store_dwordx4 %s, %s, %s offset:%d" % (vr(V_BIASOFF), vr(V_SK + 4 * mt, 4), sr(S_SRC, 2), mt * 64), ("stat", mt))
        A.vm("global_store_dwordx4 %s, %s, %s offset:%d" % (vr(V_BIASOFF), vr(V_SK + 16 + 4 * mt, 4), sr(S_SRC, 2), 1024 + mt * 64),
             ("stat", 4 + mt))
    e("s_mov_b64 exec, -1")
    A.drain()
    A.label(".L%s_epi1done" % name)
    stamp(3)
    e("s_barrier")                                   # the intermediate boards are complete
    stamp(4)
    if debug == 3:
        c("DEBUG 3: the intermediate boards (relu(conv1 + b1)) back out, stop")
        return dump_lds_and_stop()
    c("ONE convolution per block (kernarg flags bit 0): the image is the result -> y, next segment")
    e("s_bitcmp1_b32 %s, 0" % sr(S_FLAGS))
    e("s_cbranch_scc0 .L%s_twoconv" % name)
    st_regs()
    image_to_y()
    A.drain(vm=False, lg=True)
    e("s_branch .L%s_blockend" % name)
    A.label(".L%s_twoconv" % name)

    # =============================================================== conv2 + epilogue 2
    conv("c2")
    A.label(".L%s_epi2" % name)
    c("---- epilogue 2: relu(acc + b2 + skip) -> 16 bits -> Y (the next block's skip operand) and LDS (its input)")
    stamp(5)
    bias_loads(1024)
    e("s_barrier")                                   # every wave has finished reading the intermediate boards
    if debug == 5:
        A.drain()
        stamp(6)

    # The skip operand and the block's output cross the LDS image, so that every global access is a whole 16-byte piece of
    # a contiguous 1-KiB run per wave-instruction (the accumulator layout holds 8-byte pieces of 16 different pixels per
    # instruction: as direct loads / stores they cost 23 k of a block's 209 k cycles, in-kernel stamps of round 4):
    #   1. x (the block's input) -> LDS image, coalesced, like the prologue's staging (the intermediate boards are dead now)
    #   2. per accumulator tile: skip = the lane's own 8 bytes of the image; result written back to the same 8 bytes
    #   3. LDS image (= the next block's input) -> Y, coalesced; no barrier after it: the next convolution only reads LDS
    st_regs()
    for slot, base in ((0, S_SKIP0), (1, S_SKIP1)):
        e("s_mov_b64 %s, %s" % (sr(S_SRC, 2), sr(base, 2)))
        for j in range(18):
            if slot == 1:
                # (this register group still feeds board 0's LDS write j: wait for that write to have been issued ...)
                A.wait(vm_tag=("x", 0, j))
                adr, lo = image_addr(0, j)
                A.lg("ds_write_b128 %s, %s offset:%d" % (adr, vr(GROUPS[j], 4), lo), ("xw", 0, j))
            A.vm("global_load_dwordx4 %s, %s, %s" % (vr(GROUPS[j], 4), vr(ST_G), sr(S_SRC, 2)), ("x", slot, j))
            if j < 17:
                e("s_add_u32 %s, %s, 0x1000" % (sr(S_SRC), sr(S_SRC)))
                e("s_addc_u32 %s, %s, 0" % (sr(S_SRC + 1), sr(S_SRC + 1)))
    for j in range(18):
        A.wait(vm_tag=("x", 1, j))
        adr, lo = image_addr(1, j)
        A.lg("ds_write_b128 %s, %s offset:%d" % (adr, vr(GROUPS[j], 4), lo), ("xw", 1, j))
    A.wait(vm_tag=("bias", MT - 1))
    A.drain()
    e("s_barrier")                                   # the image holds x
    if debug == 5:
        stamp(8)

    AHEAD = 4                                        # skip reads in flight
    tiles = [(nt, mt) for nt in range(NT) for mt in range(MT)]

    def skip_read(i):
        nt, mt = tiles[i]
        A.lg("ds_read_b64 %s, %s offset:%d" % (vr(V_SK + (i % 8) * 2, 2), vr(V_LDSW + nt // 7), (nt % 7) * 16 * PS + mt * 32),
             ("sk", i))

    for i in range(AHEAD):
        skip_read(i)
    for i, (nt, mt) in enumerate(tiles):
        if i + AHEAD < len(tiles):
            skip_read(i + AHEAD)
        t = nt * MT + mt
        tmp = V_T + 10 * (i % 3)
        src = tile_src(t, tmp)
        e("v_pk_add_f32 %s, %s, %s" % (vr(tmp + 4, 2), vr(src, 2), vr(V_BIAS + 4 * mt, 2)))
        e("v_pk_add_f32 %s, %s, %s" % (vr(tmp + 6, 2), vr(src + 2, 2), vr(V_BIAS + 4 * mt + 2, 2)))
        A.wait(lg_tag=("sk", i))
        sk = V_SK + (i % 8) * 2
        if dt == "bf16":
            e("v_lshlrev_b32_e32 %s, 16, %s" % (vr(tmp + 0), vr(sk)))
            e("v_and_b32_e32 %s, 0xffff0000, %s" % (vr(tmp + 1), vr(sk)))
            e("v_lshlrev_b32_e32 %s, 16, %s" % (vr(tmp + 2), vr(sk + 1)))
            e("v_and_b32_e32 %s, 0xffff0000, %s" % (vr(tmp + 3), vr(sk + 1)))
        else:
            e("v_cvt_f32_f16_e32 %s, %s" % (vr(tmp + 0), vr(sk)))
            e("v_cvt_f32_f16_sdwa %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (vr(tmp + 1), vr(sk)))
            e("v_cvt_f32_f16_e32 %s, %s" % (vr(tmp + 2), vr(sk + 1)))
            e("v_cvt_f32_f16_sdwa %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (vr(tmp + 3), vr(sk + 1)))
        e("v_pk_add_f32 %s, %s, %s" % (vr(tmp + 4, 2), vr(tmp + 4, 2), vr(tmp + 0, 2)))
        e("v_pk_add_f32 %s, %s, %s" % (vr(tmp + 6, 2), vr(tmp + 6, 2), vr(tmp + 2, 2)))
        for k4 in range(4):
            e("v_max_f32_e32 %s, %s, %s" % (vr(tmp + 4 + k4), vr(V_FLOOR), vr(tmp + 4 + k4)))
        e("%s %s, %s, %s" % (cvt, vr(tmp + 8), vr(tmp + 4), vr(tmp + 5)))
        e("%s %s, %s, %s" % (cvt, vr(tmp + 9), vr(tmp + 6), vr(tmp + 7)))
        A.lg("ds_write_b64 %s, %s offset:%d" % (vr(V_LDSW + nt // 7), vr(tmp + 8, 2), (nt % 7) * 16 * PS + mt * 32), ("w2", t))
        # (the temporaries of this tile are rewritten three tiles on: by then its LDS write has long been issued)
    A.drain()
    e("s_barrier")                                   # the image holds the block's output = the next block's input
    if debug == 5:
        stamp(9)
    image_to_y()
    A.drain(vm=False, lg=True)
    if debug == 5:
        A.drain()
    stamp(7)
    if debug == 5:
        e("s_add_u32 s52, s52, 128")
        e("s_addc_u32 s53, s53, 0")
    A.label(".L%s_blockend" % name)
    c("next block: its skip operand is what was just stored")
    e("s_mov_b64 %s, %s" % (sr(S_SKIP0, 2), sr(S_Y0, 2)))
    e("s_mov_b64 %s, %s" % (sr(S_SKIP1, 2), sr(S_Y1, 2)))
    e("s_add_u32 %s, %s, 0x800" % (sr(S_BP), sr(S_BP)))
    e("s_addc_u32 %s, %s, 0" % (sr(S_BP + 1), sr(S_BP + 1)))
    e("s_sub_u32 %s, %s, 1" % (sr(S_BLK), sr(S_BLK)))
    e("s_cmp_lg_u32 %s, 0" % sr(S_BLK))
    e("s_cbranch_scc1 .L%s_block" % name)
    c("a head hands its pair over: every wave's stores done, then one wave writes the L2 back and raises the pair's flag")
    e("s_bitcmp1_b32 %s, 1" % sr(S_MODE))
    e("s_cbranch_scc0 .L%s_next" % name)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")
    e("s_cmp_lg_u32 %s, 0" % sr(S_WAVE))
    e("s_cbranch_scc1 .L%s_next" % name)
    e("s_lshl_b32 %s, %s, 2" % (sr(S_T1), sr(S_PAIR)))
    e("s_add_u32 %s, %s, 0x%x" % (sr(S_FLAG), sr(S_PLAN), MAX_WG * PLAN_STRIDE))
    e("s_addc_u32 %s, %s, 0" % (sr(S_FLAG + 1), sr(S_PLAN + 1)))
    e("s_add_u32 %s, %s, %s" % (sr(S_FLAG), sr(S_FLAG), sr(S_T1)))
    e("s_addc_u32 %s, %s, 0" % (sr(S_FLAG + 1), sr(S_FLAG + 1)))
    e("buffer_wbl2 sc1")
    e("s_waitcnt vmcnt(0)")
    e("v_mov_b32_e32 %s, 0" % vr(V_TMP))
    e("v_mov_b32_e32 %s, %s" % (vr(V_TMP + 1), sr(S_B1)))
    e("global_store_dword %s, %s, %s sc1" % (vr(V_TMP), vr(V_TMP + 1), sr(S_FLAG, 2)))
    e("s_waitcnt vmcnt(0)")
    e("s_branch .L%s_next" % name)
    A.label(".L%s_end" % name)
    e("s_endpgm")
    return A.lines


def kernel_text(name, dt, debug=0):
    body = gen_kernel(name, dt, debug)
    out = ["\t.text", "\t.globl\t%s" % name, "\t.p2align\t8", "\t.type\t%s,@function" % name, "%s:" % name]
    out += body
    out += [".L%s_fend:" % name, "\t.size\t%s, .L%s_fend-%s" % (name, name, name), ""]
    out += ["\t.section\t.rodata,\"a\",@progbits", "\t.p2align\t6, 0x0", "\t.amdhsa_kernel %s" % name,
            "\t\t.amdhsa_group_segment_fixed_size %d" % LDS_BYTES,
            "\t\t.amdhsa_private_segment_fixed_size 0",
            "\t\t.amdhsa_kernarg_size 80",
            "\t\t.amdhsa_user_sgpr_count 2",
            "\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1",
            "\t\t.amdhsa_system_sgpr_workgroup_id_x 1",
            "\t\t.amdhsa_system_sgpr_workgroup_id_y 0",
            "\t\t.amdhsa_system_sgpr_workgroup_id_z 0",
            "\t\t.amdhsa_system_vgpr_workitem_id 0",
            "\t\t.amdhsa_next_free_vgpr %d" % (NEXT_VGPR + 256),
            "\t\t.amdhsa_next_free_sgpr %d" % NEXT_SGPR,
            "\t\t.amdhsa_accum_offset %d" % NEXT_VGPR,
            "\t\t.amdhsa_reserve_vcc 1",
            "\t\t.amdhsa_float_round_mode_32 0",
            "\t\t.amdhsa_float_round_mode_16_64 0",
            "\t\t.amdhsa_float_denorm_mode_32 3",
            "\t\t.amdhsa_float_denorm_mode_16_64 3",
            "\t\t.amdhsa_dx10_clamp 1",
            "\t\t.amdhsa_ieee_mode 1",
            "\t.end_amdhsa_kernel", ""]
    return out


def metadata(names):
    out = ["\t.amdgpu_metadata", "---", "amdhsa.kernels:"]
    for name in names:
        out += ["  - .agpr_count:     256", "    .args:"]
        for i in range(6):
            out += ["      - .address_space:  global", "        .offset:         %d" % (8 * i), "        .size:           8",
                    "        .value_kind:     global_buffer"]
        for off in (48, 52):
            out += ["      - .offset:         %d" % off, "        .size:           4", "        .value_kind:     by_value"]
        out += ["      - .address_space:  global", "        .offset:         56", "        .size:           8",
                "        .value_kind:     global_buffer"]
        out += ["      - .offset:         64", "        .size:           4", "        .value_kind:     by_value"]
        out += ["      - .address_space:  global", "        .offset:         72", "        .size:           8",
                "        .value_kind:     global_buffer"]
        out += ["    .group_segment_fixed_size: %d" % LDS_BYTES, "    .kernarg_segment_align: 8", "    .kernarg_segment_size: 80",
                "    .max_flat_workgroup_size: 256", "    .name:           %s" % name, "    .private_segment_fixed_size: 0",
                "    .sgpr_count:     %d" % (NEXT_SGPR + 6), "    .sgpr_spill_count: 0", "    .symbol:         %s.kd" % name,
                "    .uniform_work_group_size: 1", "    .uses_dynamic_stack: false", "    .vgpr_count:     %d" % (NEXT_VGPR + 256),
                "    .vgpr_spill_count: 0", "    .wavefront_size: 64"]
    out += ["amdhsa.target:   amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", "", "\t.end_amdgpu_metadata"]
    return out


def main():
    path = sys.argv[1]
    debug = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    out = ["; generated by gen_tower_asm.py -- do not edit", "\t.amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"",
           "\t.amdhsa_code_object_version 6", ""]
    names = ["hive_tower72_bf16", "hive_tower72_f16"]
    for name, dt in zip(names, ("bf16", "f16")):
        out += kernel_text(name, dt, debug)
    out += ["\t.section\t.rodata,\"a\",@progbits", "\t.p2align\t8, 0x0", "\t.type\thive_tap_table,@object",
            "hive_tap_table:"]
    tab = tap_table()
    for i in range(0, len(tab), 16):
        out.append("\t.long\t" + ", ".join(str(v) for v in tab[i:i + 16]))
    out += ["\t.size\thive_tap_table, %d" % (4 * len(tab)), ""]
    out += metadata(names)
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
