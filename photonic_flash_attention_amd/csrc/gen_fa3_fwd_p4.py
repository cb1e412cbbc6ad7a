#!/usr/bin/env python3
"""gen_fa3_fwd_p4.py -- generator of the PERSISTENT 4-wave Flash-Attention forward for gfx950, as assembly.

    python3 gen_fa3_fwd_p4.py > build/fa3_fwd_p4.s        (the Makefile assembles it into the code object that libpfa_hip.so embeds)

Replaces the reference's tile loop (core/flash_attention_3.py:207-260) on the shapes pfa_capi.hip routes here (pfa_p4.hip
p4_eligible: D = 128 or 64, Sq >= 128, Sk >= 193, no element mask): the same math, LDS images and MFMA operand maps as
fa3_fwd_w4_kernel.h (4 waves x 64 query rows, one wave per SIMD, S^T = K Q^T and O^T += V^T P^T on v_mfma_f32_32x32x16, online
softmax with defer-max), but

  * PERSISTENT: one workgroup per CU walks a static list of (head, Q block) items; the K/V LDS-DMA ring keeps running across
    the item seam (the last two iterations of an item already fetch tiles 0 / 1 of the next one), the next item's Q rows are
    fetched into their landing zone during the current item's first four tiles, the output leaves through LDS as whole rows.
    Under a causal mask a unit is the (heaviest, lightest) open block pair of a head, so the static shares are equal; the units
    of a head are dealt to consecutive CUs of ONE XCD, which therefore stream that head's K/V in lockstep (L2 serves all but one).
  * every instruction of the tile loop is placed by this script: one MFMA per gap plus its fillers, LDS reads waited for in
    pairs (8 s_waitcnt per phase instead of 16), one M0 write per four DMA pieces (a wave's pieces are contiguous in LDS, the
    piece index rides in the instruction's immediate offset), the tile offset in the load's scalar offset (no VALU), no
    compiler-inserted s_nop / v_mov; the rare paths (O rescale, diagonal mask, a wave's last tile) sit out of line.
  * the tile loop exists twice (generic iteration / lean double iteration far from an item's ends), an item's last iteration runs on
    its successor's Q fragments (body_seam), and three flavours share the code: plain (whole blocks and tiles), *_km_* ([B, Sk] key
    mask: the waves read the mask bytes themselves) and *_kl_* (ragged lengths, seqlens_k); all of that is described where it is
    emitted (kernel(), body_seam(), mask_keys(), Gen.__init__).

Register plan (per wave, 512 registers; D = 128 -- D = 64 halves O and Q):
    a[0:63] O of strip A, a[64:127] O of strip B, a[128:159] / a[160:191] the Q fragments of A / B, a[192:255] the K / V^T fragment rings
    v[16:143] S, double buffered: buf0 A, buf0 B, buf1 A, buf1 B (32 each: key block 0, key block 1)
    v[144:175] P (packed 16-bit) of A / B, v[176:207] the low halves of a split P, v[208:] addresses and state, v[0:15] scratch
"""
import os
import sys


# Experiment / diagnostic knobs (P4_*): read from the environment ONLY with P4_DEV=1 (tools/p4_variants.py builds such code objects for
# same-process A/B runs and stamped diagnostics; they are loaded by tools/p4_ab.py, never by the library).  The product build
# (`make`) runs without P4_DEV: a P4_* knob left in the environment then stops the build instead of silently linking a stamped or
# ablated code object into libpfa_hip.so.
DEV = os.environ.get("P4_DEV", "") == "1"
if not DEV:
    _stray = sorted(k for k in os.environ if k.startswith("P4_"))
    if _stray:
        sys.exit(f"gen_fa3_fwd_p4.py: {', '.join(_stray)} set without P4_DEV=1 -- the product code object takes no experiment knobs")


def knob(name, default):
    return os.environ.get(name, default) if DEV else default

# ------------------------------------------------------------------------------------------------------------------------------
# LDS map (bytes): K ring 2 x 16 KiB, V ring 2 x 16 KiB, Q landing zone 4 waves x 16 KiB, O staging 4 waves x 8 KiB = 160 KiB
# per kernel (Gen.__init__): TILE = 64 keys x D x 2 B (16 KiB at D = 128, 8 KiB at D = 64); K ring at 0, V ring at 2 TILE, the waves'
# Q landing zones (64 rows each = TILE) at 4 TILE, their O staging (32 rows = TILE / 2) at 8 TILE: 10 TILE in all

# kernarg dwords (struct pfa::P4Params in pfa_p4.hip -- static_asserted there against these offsets)
KA = dict(q=0, k=2, v=4, o=6, lse=8, q_sb=10, q_sh=11, k_sb=12, k_sh=13, v_sb=14, v_sh=15, o_sb=16, o_sh=17,
          q_ss=18, k_ss=19, v_ss=20, o_ss=21, H=22, Sq=23, Sk=24, NB=25, NU=26, magic_NU=27, magic_H=28, kv_group=29,
          magic_G=30, scale_log2=31, thr=32, hx=33, xcd_mode=34, SL=35, nt_full=36, pad=37, dbg=38)
KARG_DWORDS = 40                    # preloaded into s[60:99]
KARG_MEM_DWORDS = 42                # ... of the kernarg block's 42: dwords 40 / 41 = seqlens_k, fetched where it is needed (decode)
KA_SEQLENS = 40
KBASE_SGPR = 60                     # kernargs live in s[60:99]


def ka(name, n=1, hi=False):
    i = KBASE_SGPR + KA[name] + (1 if hi else 0)
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


class SAlloc:
    def __init__(self, first, last):
        self.next, self.last, self.names = first, last, {}

    def new(self, name, n=1, align=1):
        while self.next % align:
            self.next += 1
        base = self.next
        self.next += n
        assert self.next - 1 <= self.last, f"out of SGPRs at {name}"
        self.names[name] = (base, n)
        return base

    def __call__(self, name, i=None, n=None):
        base, cnt = self.names[name]
        if i is not None:
            return f"s{base + i}" if n is None else f"s[{base + i}:{base + i + n - 1}]"
        return f"s{base}" if cnt == 1 else f"s[{base}:{base + cnt - 1}]"


S = SAlloc(3, KBASE_SGPR - 1)
for nm, n, al in [("wave", 1, 1), ("ksrd", 4, 4), ("vsrd", 4, 4), ("qsrd_n", 4, 4), ("osrd", 4, 4), ("lsrd", 4, 4),
                  ("ksrd_n", 2, 2), ("vsrd_n", 2, 2), ("grow", 2, 2), ("t2", 2, 2),
                  ("koff", 1, 1), ("voff", 1, 1), ("ktile", 1, 1), ("vtile", 1, 1), ("krem", 1, 1), ("vrem", 1, 1),
                  ("qrem", 1, 1), ("wrem", 1, 1), ("irem", 1, 1), ("nt_n", 1, 1), ("qdst", 1, 1), ("qoff", 1, 1),
                  ("t0", 1, 1), ("t1", 1, 1), ("t3", 1, 1), ("w4k", 1, 1), ("nd_n", 1, 1), ("nd", 1, 1), ("L_n", 1, 1),
                  ("n_u", 1, 1), ("n_sub", 1, 1), ("n_valid", 1, 1), ("n_qblk", 1, 1), ("n_b", 1, 1), ("n_hh", 1, 1), ("n_nt", 1, 1)]:
    S.new(nm, n, al)


# VGPR plan
def SBUF(buf, X, kb, e):            # S accumulator register of buffer buf (0/1), strip X ('A'/'B'), key block kb, element e
    return 16 + 64 * buf + (0 if X == 'A' else 32) + 16 * kb + e


def PD(X, i):
    return 144 + (0 if X == 'A' else 16) + i


def PD0B(X, i):             # fast loop: second buffer of P dwords 0..7 (key block 0) for tiles in S buffer 1 -- they are written while the
    return 176 + (0 if X == 'A' else 8) + i    # PV product of the tile before still reads the first one  (v[176:191]: a split P's low halves otherwise)


FT = {'A': [0, 1, 2, 3], 'B': [8, 9, 10, 11]}      # fast loop: where the exponentials land before they are summed and packed (V_E scratch)


def PS1(X):                 # fast loop: row sum of key block 1, per strip (folded into l one phase later)
    return 248 if X == 'A' else 249


def PDL(X, i):              # split P: the low halves (the VGPRs the 4-deep fragment rings would use)
    return 176 + (0 if X == 'A' else 16) + i


PKADD = int(knob("P4_PKADD", "0"))        # EXPERIMENT, off: row sums as v_pk_add_f32 on register pairs (32 instead of 64 instructions a tile) ran 5-7 % SLOWER
SEAM = int(knob("P4_SEAM", "1"))          # the item seam as one more FULL iteration (kernel(), body_seam): 0 = LAST body, epilogue, prologue one after the other
ILV = int(knob("P4_ILV", "1"))            # the FULL body's softmax streams of strips A and B interleaved instruction by instruction
WAITN = int(knob("P4_WAITN", "2"))        # LDS fragments waited for at a time (2: 8 s_waitcnt per phase; 4: 4)
LEAN = int(knob("P4_LEAN", "1"))          # the tile loop's lean path (kernel()): 0 = every iteration carries the full bookkeeping
RING = int(knob("P4_RING", "8"))          # K / V^T fragment rings: 4 (VGPRs) or 8 (the spare accumulator registers a[192:255])
KFR = lambda i: (192 + 4 * (i % 8)) if RING == 8 else (176 + 4 * (i % 4))
VFR = lambda i: (224 + 4 * (i % 8)) if RING == 8 else (192 + 4 * (i % 4))
KOFF = lambda ks: 208 + ks
VOFF = lambda sel, hi: 216 + 2 * sel + hi       # sel: d block (D = 128) / 2 * (k-step parity) + d block (D = 64)
KDOFF = lambda t: 224 + t
VDOFF = lambda t: 228 + t
QDOFF = lambda t: 232 + t
ST = {"m": 0, "l": 1, "mc": 2, "thr": 3, "al": 4, "ps0": 5}


def STV(X, f):
    return 236 + (0 if X == 'A' else 6) + ST[f]


def SV(X, f):               # row sum / scaled max of the item that is ending, parked in S buffer 1 across the pipelined seam (body_seam)
    return 80 + (0 if X == 'A' else 2) + (0 if f == 'l' else 1)


def PSP(X):                 # packed row-sum accumulator of a strip (two lanes of a v_pk_add_f32): v[176:177] / v[178:179]
    return 176 + (0 if X == 'A' else 2)


V_PS1 = 248
V_MX = 249
V_T = [250, 251, 252, 253]
V_TA = 254            # causal: r + 1 - 4h (element-mask threshold of the diagonal tile, see mask_diag)
V_LANE = 255
V_E = list(range(0, 16))   # prologue / epilogue / rescale scratch (v0 = workitem id at entry)
STAMP = int(knob("P4_STAMP", "0"))      # 1: every phase; 2: one stamp per iteration only (buckets 0 / 1 stay empty); 3: kernel totals only (lean loop and pipelined seam stay on)
LIGHT1 = int(knob("P4_LIGHT1", "0"))      # EXPERIMENT: causal units run their light block first (the heavy streams of a head then follow each other within 28 tiles)
MB = int(knob("P4_MB", "1"))              # fast loop: the iteration's barrier sits INSIDE the PV phase (behind the last V^T read), and the first K fragments of the
                                          # next QK^T phase are requested right behind it -- their latency and the barrier skew run under the rest of PV(j)
MBGAP = int(knob("P4_MBGAP", "0"))        # ... behind this MFMA gap of the PV phase (0: 27 of 32 at D = 128, 12 of 16 at D = 64 -- same box, gaps 16..27 / 8..13:
                                          # the late barrier is worth +0.4..1 % over gap 20 at D = 128; at D = 64 gap 13 loses 2 %)
PSTART = int(knob("P4_PSTART", "1"))      # parity variant: key block 0's P is packed by the START stream, in the second half of the PV phase (behind the MFMAs
                                          # that still read the previous tile's dwords 0..7), not by the finish: the QK^T phase carried 330 instructions in 32 gaps
KMFAST = int(knob("P4_KMFAST", "1"))      # key-mask kernels on the fast loop too (fresh rows: see Gen.__init__)
DIET = int(knob("P4_DIET", "1"))          # fast loop: block sums start with t0 + t1 (no zeroing), one compare + s_cbranch_vccnz per tile for both strips
FASTMAX = int(knob("P4_FASTMAX", "1"))    # the tile loop without a row max (Gen.fast; see finish_fast): a tile's exponentials are taken against the running
                                          # maximum, its scaled scores stay in the S buffer, and a row sum past 2^14 sends the wave to a fix-up subroutine
PRE = int(knob("P4_PRE", "24"))            # issue cycles of the finish stream placed between the first K-fragment reads and the first QK^T MFMA (their latency)
STAMP_BASE = 192 if FASTMAX else 184   # (the fast loop's second P buffer lives in v[176:191])
_STAMP_BASE_DOC = 184                          # v[184:199]: previous clock, buckets 0..14 (free in the fast variant: the low halves of a split P live there)
ABL = knob("P4_ABL", "")                                         # timing-only ablations, see dma_plan
DMA_PRICE = int(knob("P4_DMA_PRICE", "30"))                      # issue cycles budgeted for one LDS-DMA piece
DMA_GAPS_V = [int(x) for x in knob("P4_DMA_GAPS_V", "1,5,9,13").split(",")]      # QK^T gaps that carry the V(j+1) pieces
DMA_GAPS_K = [int(x) for x in knob("P4_DMA_GAPS_K", "17,21,25,27").split(",")]   # ... and the K(j+2) pieces
NODIAG = "0x40000000"      # causal ragged kernels, nd_n / nd: the item has no diagonal tile (it lies behind its batch's seqlens_k cut)
NEG_BIG = "0xf149f2ca"     # -1e30f
NEG_INF = "0xff800000"


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=1):
    return f"a{base}" if n == 1 else f"a[{base}:{base + n - 1}]"


def fr(base, n):            # a fragment-ring register range: accumulator file when the rings are 8 deep
    return ar(base, n) if RING == 8 else vr(base, n)


class Lgkm:
    """Outstanding LDS reads in issue order -> counted s_waitcnt lgkmcnt(N)."""

    def __init__(self):
        self.q = []

    def issue(self, tag):
        self.q.append(tag)

    def need(self, tags):
        idx = [self.q.index(t) for t in tags if t in self.q]
        if not idx:
            return None
        i = max(idx)
        n = len(self.q) - 1 - i
        self.q = self.q[i + 1:]
        return f"s_waitcnt lgkmcnt({n})"


class Gen:
    def __init__(self, dtype, causal, out32=False, split=False, D=128, kmask=False, klen=False):
        """out32: fp32 store (straight from the accumulators); split: P enters the PV product as a 16-bit hi + lo pair (two MFMAs per
        fragment, P's rounding error 2^-18 instead of 2^-9): together the <= 1e-3 parity variant on the benched schedule."""
        assert out32 == split, "the code object carries the fast variant (16-bit store, one P) and the parity variant (fp32 store, split P)"
        self.dt, self.causal, self.out32, self.split, self.kmask = dtype, causal, out32, split, kmask
        # klen: RAGGED problems -- Sq / Sk are no multiples of the block / tile.  Rows past the end are kept out by the buffer
        # descriptors (K / V: the whole slab as ever; Q / O / LSE: records per item = the block's existing rows; a raw buffer's range
        # check is offset >= records - scalar offset, so the tile / group / row-group offsets riding in scalar offsets are covered:
        # loads past the end give zeros, stores past it are dropped), keys past Sk by a mask word computed from the length (not
        # needed under the causal mask: a valid row never looks that far).
        self.klen = klen
        assert not (kmask and klen)
        # key-mask kernels: the keys left from the tile whose mask bytes are fetched next ride in a register the flavour does not use
        # otherwise -- the causal kernels' sub-item flag without the causal mask, the seqlens register (non-causal only) with it --
        # and start an item at the keys its batch has (seqlens_k; Sk without it, and always under the causal mask)
        self.kleft, self.kleft0 = (S('L_n'), ka('Sk')) if causal else (S('n_sub'), S('L_n'))
        self.mwords = kmask or klen            # (causal ragged kernels: the words carry seqlens_k; without it they are all ones where a row can look)
        self.nd_n, self.nd = S('nd_n'), S('nd')
        # ... `Lcut`: where decode leaves the next item's key count L -- ragged kernels: L_n (it feeds their length words); key-mask kernels:
        # the nt_full kernarg slot (unused under the causal mask), which is their `kleft0`, the keys the mask-byte stream starts an item with
        self.Lcut = S('L_n') if klen else ka('nt_full')
        if kmask and causal:
            self.kleft0 = ka('nt_full')
        self.cl = (klen or kmask) and causal              # seqlens_k under the causal mask: an item's tile count is cut to its batch's keys (decode)
        # fast: the plain kernels of the fast variant -- every row sees a key in its item's tile 0 (so its maximum is finite from there on)
        # (ragged kernels too: with a prefix of visible keys -- Sk, seqlens_k -- a row sees key 0 unless its batch has no key at all, and then
        #  it sees none in any tile; start_fast clamps such a row's maximum to -1e30, its weights are exp2(-inf) = 0.  Key-mask kernels: a
        #  row may come alive in a later tile (left padding).  Such a FRESH row keeps m c = 0, so that x = s c holds the score itself, and a
        #  negative limit in STV thr: the tile check then fires for it whatever its sums are (unless the tile's mask word is zero) and the
        #  fix-up gives it its first maximum.)
        self.fast = bool(FASTMAX) and not split and (not kmask or bool(KMFAST))
        self.ret = S('grow') if DIET else "s[58:59]"          # fix-up subroutine's return address (s[58:59] is a mask word of the ragged kernels)
        self.mb = self.fast and bool(MB) and STAMP in (0, 3)      # (on the parity variant's SAFE bodies it buys nothing at D = 128 and costs 4 % at D = 64)
        assert not (kmask and STAMP), "the key-mask kernels keep their mask words where the stamps keep their clock (s[58:59], the dbg kernarg)"
        assert D in (64, 128)
        self.D, self.KS, self.DB = D, D // 16, D // 32                 # head dim, k-steps of QK^T, 32-wide d blocks of PV
        self.NKF, self.NVF = 2 * self.KS, 4 * self.DB                   # K / V^T fragments per 64-key tile
        self.TILE = 128 * D
        self.HALF = self.TILE // 2
        self.K_BASE, self.V_BASE, self.Q_BASE, self.O_BASE = 0, 2 * self.TILE, 4 * self.TILE, 8 * self.TILE
        self.LDS_BYTES = 10 * self.TILE
        self.PPW = self.TILE // 4096                                    # 1-KiB DMA pieces per wave and image
        self.RB, self.CPR = 2 * D, D // 8                               # bytes and 16-byte chunks of a Q / O row
        self.RING = min(RING, self.NKF)
        self.mf = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.name = f"fa3_fwd_p4_{dtype}_d{D}_{'causal' if causal else 'full'}{'_km' if kmask else ('_kl' if klen else '')}_{'splitp_o32' if out32 else 'o16'}"
        self.main, self.ool, self.ool2, self.lstack = [], [], [], []
        self.L = self.main
        self.abl_on = False
        self.uid = 0

    def OA(self, X, db, e=0):                  # accumulator register of O, strip X, d block db, element e
        return (0 if X == 'A' else 16 * self.DB) + 16 * db + e

    def QA(self, X, ks):                       # accumulator registers of the Q fragment of k-step ks
        return 32 * self.DB + (0 if X == 'A' else 4 * self.KS) + 4 * ks

    # ---- emission helpers ------------------------------------------------------------------------------------------------
    def i(self, s):
        if self.abl_on and ABL:            # TIMING-ONLY ablations of the FULL body (wrong results): drop one instruction class
            op = s.split()[0]
            drop = {"noadd": op == "v_add_f32", "noexp": op == "v_exp_f32", "nofma": op == "v_fma_f32", "nocvt": op.startswith("v_cvt_pk"),
                    "nomax": op.startswith("v_max"), "nodma2": op.startswith("buffer_load"), "nolds": op.startswith("ds_read") or s.startswith("s_waitcnt lgkmcnt"),
                    "noqk": op.startswith("v_mfma") and s.split()[1].startswith("v["), "nopv": op.startswith("v_mfma") and s.split()[1].startswith("a["),
                    "nosoftmax": op in ("v_add_f32", "v_exp_f32", "v_fma_f32", "v_max3_f32", "v_max_f32", "v_mul_f32", "v_sub_f32", "v_mov_b32", "v_cndmask_b32",
                                        "v_cmp_gt_f32", "v_permlane32_swap_b32", "s_nop", "s_or_b64") or op.startswith("v_cvt_pk")}
            if any(drop.get(a, False) for a in ABL.split("+")):
                return
        self.L.append("\t" + s)

    def lab(self, s):
        self.L.append(f"{s}:")

    def cm(self, s):
        self.L.append(f"\t; {s}")

    def ul(self, stem):
        self.uid += 1
        return f".L{self.name}_{stem}_{self.uid}"

    def emit(self, lst):
        for x in lst:
            if x.startswith("@"):              # a packer marker (pack()): nothing to emit
                continue
            if x.endswith(":"):
                self.lab(x[:-1])
            else:
                self.i(x)

    def out_of_line(self, on):
        """Rare paths are emitted behind the kernel's s_endpgm: on = switch to that region (from inside it: to a second one behind it,
        so that a rare path may have rare paths of its own), off = back to where the matching `on` came from."""
        if on:
            self.lstack.append(self.L)
            self.L = self.ool if self.L is self.main else self.ool2
        else:
            self.L = self.lstack.pop()

    # ---- diagnostic stamps (P4_STAMP=1 at generation; never time such a build) -------------------------------------------------------
    # a192 = previous s_memtime (low word), a[193 + k] = cycles accumulated in bucket k: 0 QK^T phase, 1 PV phase, 2 wait + barrier +
    # stream bookkeeping, 3 item switch + prologue, 4 epilogue, 5 LAST bodies, 6 SKIP bodies, 7 FULL iterations (count), 8 items (count)
    def stamp(self, k, count=None, fine=False):
        if not STAMP or STAMP == 3 or (fine and STAMP == 2):
            return
        assert RING == 8, "stamps live in the VGPRs the 4-deep rings use"
        if self.split:
            return                             # ... and so do the low halves of a split P: the parity variant carries no stamps
        t2 = vr(V_E[14])
        self.i("s_memtime s[58:59]")
        self.i("s_waitcnt lgkmcnt(0)")
        SB = STAMP_BASE
        self.i(f"v_sub_u32 {t2}, s58, v{SB}")
        self.i(f"v_add_u32 v{SB + 1 + k}, v{SB + 1 + k}, {t2}")
        self.i(f"v_mov_b32 v{SB}, s58")
        if count is not None:
            self.i(f"v_add_u32 v{SB + 1 + count}, 1, v{SB + 1 + count}")

    def stamp_init(self):
        if not STAMP:
            return
        SB = STAMP_BASE
        for k in range(SB + 1, SB + 16):
            self.i(f"v_mov_b32 v{k}, 0")
        self.i("s_memtime s[58:59]")
        self.i("s_waitcnt lgkmcnt(0)")
        self.i(f"v_mov_b32 v{SB}, s58")
        self.i(f"v_mov_b32 v{SB + 11}, s58")                         # bucket 10: cycle counter at kernel start
        self.i("s_memrealtime s[58:59]")
        self.i("s_waitcnt lgkmcnt(0)")
        self.i(f"v_mov_b32 v{SB + 12}, s58")                         # bucket 11: 100 MHz counter at kernel start
        self.i("s_getreg_b32 s58, hwreg(HW_REG_XCC_ID)")             # bucket 13: the XCD this workgroup really runs on (bits 3:0)
        self.i(f"v_mov_b32 v{SB + 14}, s58")
        self.i("s_getreg_b32 s58, hwreg(HW_REG_HW_ID)")              # bucket 14: CU / SH / SE ids
        self.i(f"v_mov_b32 v{SB + 15}, s58")

    def stamp_dump(self):
        """[workgroup][wave][16] dwords into the dbg buffer (kernarg), by lane 0."""
        if not STAMP:
            return
        t3 = vr(V_E[15])
        lskip = self.ul("nodbg")
        self.i(f"s_cmp_eq_u64 {ka('dbg', 2)}, 0")
        self.i(f"s_cbranch_scc1 {lskip}")
        self.i("s_memtime s[58:59]")
        self.i("s_waitcnt lgkmcnt(0)")
        SB = STAMP_BASE
        self.i(f"v_sub_u32 v{SB + 11}, s58, v{SB + 11}")             # total cycles
        self.i("s_memrealtime s[58:59]")
        self.i("s_waitcnt lgkmcnt(0)")
        self.i(f"v_sub_u32 v{SB + 12}, s58, v{SB + 12}")             # total 10-ns ticks
        self.i(f"s_mov_b32 {S('lsrd', 0)}, {ka('dbg')}")
        self.i(f"s_and_b32 {S('lsrd', 1)}, {ka('dbg', hi=True)}, 0xffff")
        self.i(f"s_mov_b32 {S('lsrd', 2)}, 0x7fffffff")
        self.i(f"s_lshl_b32 {S('t0')}, s2, 2")
        self.i(f"s_add_u32 {S('t0')}, {S('t0')}, {S('wave')}")
        self.i(f"s_lshl_b32 {S('t0')}, {S('t0')}, 6")                # ((wg * 4 + wave) * 16) * 4 bytes
        self.i(f"v_mov_b32 {t3}, 0")
        self.i("s_mov_b64 exec, 1")
        for k in range(16):
            self.i(f"buffer_store_dword v{STAMP_BASE + k}, {t3}, {S('lsrd')}, {S('t0')} offen offset:{4 * k}")
        self.i("s_mov_b64 exec, -1")
        self.lab(lskip)

    # ---- key mask (kmask kernels): the caller's [B, Sk] bytes (contiguous rows, non-zero = visible) at the `dbg` kernarg.  Every wave
    # reads the 64 bytes of a tile itself (lane i: key 64 t + i, one global_load_ubyte) two tiles ahead, beside the K pieces of that
    # tile and with the same stream switch; at the bottom of the iteration (behind its counted vmcnt) a compare turns them into the
    # 64-bit word MK(t & 1): bit i = key 64 t + i visible.  `pad` = the running byte offset b * Sk + 64 t.
    KM_V = [V_E[14], V_E[15]]                  # the bytes in flight (scratch registers nothing else touches inside the tile loop)

    @staticmethod
    def MK(i):
        return "s[58:59]" if i == 0 else "s[100:101]"

    def len_word(self, i, dst=None, left=None):
        """dst (default MK(i)) = the low min(64, max(0, left)) bits; left (default `pad`: klen, no causal mask) = keys left from the
        tile in question on."""
        t0 = S('t0')
        dst, left = dst or self.MK(i), left or ka('pad')
        self.i(f"s_max_i32 {t0}, {left}, 0")
        self.i(f"s_min_i32 {t0}, {t0}, 63")
        self.i(f"s_bfm_b64 {dst}, {t0}, 0")
        self.i(f"s_cmp_ge_i32 {left}, 64")
        self.i(f"s_cselect_b64 {dst}, -1, {dst}")

    def km_len_and(self, i):
        """key-mask kernels without the causal mask: the word just made from the mask bytes AND the keys its batch has (seqlens_k;
        Sk without it, and under the causal mask).  The keys left: self.kleft."""
        if not self.kmask:
            return
        self.len_word(i, dst=S('t2'), left=self.kleft)
        self.i(f"s_and_b64 {self.MK(i)}, {self.MK(i)}, {S('t2')}")
        self.i(f"s_sub_u32 {self.kleft}, {self.kleft}, 64")

    def mask_load(self, n=0):
        v = vr(self.KM_V[n])
        return [f"v_add_u32 {v}, {ka('pad')}, {vr(V_LANE)}", f"global_load_ubyte {v}, {v}, {ka('dbg', 2)}"]

    def mask_word(self, i, n=0):
        return f"v_cmp_ne_u32_e64 {self.MK(i)}, 0, {vr(self.KM_V[n])}"

    def mask_keys(self, buf, i):
        """S of buffer `buf` (both strips) under the mask word MK(i): nothing to do when all 64 keys are visible (the common tile of a
        padding mask), else -inf per masked key: key 32 kb + kidx(e) + 4 h of register e -- bit b for the lower lane half, b + 4 for the upper."""
        if not self.mwords:
            return
        lm, lr = self.ul("kmask"), self.ul("kmasked")
        self.i(f"s_cmp_eq_u64 {self.MK(i)}, -1")
        self.i(f"s_cbranch_scc0 {lm}")
        self.lab(lr)
        self.out_of_line(True)
        self.lab(lm)
        self.i("s_nop 7")
        self.i("s_nop 7")                     # last QK^T MFMA -> VALU access of S
        ninf = vr(V_T[3])
        self.i(f"v_mov_b32 {ninf}, {NEG_INF}")
        for kb in range(2):
            for e in range(16):
                b = 32 * kb + (e & 3) + 8 * (e >> 2)
                self.i(f"s_bitcmp1_b64 {self.MK(i)}, {b}")
                self.i("s_cselect_b32 vcc_lo, -1, 0")
                self.i(f"s_bitcmp1_b64 {self.MK(i)}, {b + 4}")
                self.i("s_cselect_b32 vcc_hi, -1, 0")
                for X in "AB":
                    self.i(f"v_cndmask_b32 {vr(SBUF(buf, X, kb, e))}, {ninf}, {vr(SBUF(buf, X, kb, e))}, vcc")
        self.i(f"s_branch {lr}")
        self.out_of_line(False)

    # ---- building blocks -------------------------------------------------------------------------------------------------
    def kread(self, slot, i):                 # K fragment i = (kb, ks) of the K tile in ring slot `slot` -> ring register i % 4
        kb, ks = i // self.KS, i % self.KS
        return f"ds_read_b128 {fr(KFR(i), 4)}, {vr(KOFF(ks))} offset:{self.K_BASE + slot * self.TILE + kb * self.HALF}"

    def vread(self, slot, idx):               # V^T fragment idx = (f, db): two transposed 8-byte reads (rows +0 / +8)
        f, db = idx // self.DB, idx % self.DB              # k-step f = (key block, 16-key half), d block db
        if self.D == 128:                                  # one key per LDS row: the 16-key half is an offset
            ko, sel = slot * self.TILE + (f // 2) * self.HALF + (f & 1) * 16 * 256, db
        else:                                              # two keys per LDS row: the half enters the swizzle, so it has its own address registers
            ko, sel = slot * self.TILE + (f // 2) * self.HALF, 2 * (f & 1) + db
        b = VFR(idx)
        return [f"ds_read_b64_tr_b16 {fr(b, 2)}, {vr(VOFF(sel, 0))} offset:{ko}",
                f"ds_read_b64_tr_b16 {fr(b + 2, 2)}, {vr(VOFF(sel, 1))} offset:{ko}"]

    def dma(self, which, t):                  # piece t of this wave's four (M0 already points at the wave's 4 KiB of the slot)
        off, srd, so = (KDOFF(t), S("ksrd"), S("koff")) if which == 'K' else (VDOFF(t), S("vsrd"), S("voff"))
        return f"buffer_load_dwordx4 {vr(off)}, {srd}, {so} offen offset:{1024 * t} lds"

    def setm0(self, which, slot):             # M0 = this wave's 4 KiB of ring slot `slot` of the K / V ring  (clobbers SCC)
        return f"s_add_u32 m0, {S('w4k')}, {(self.K_BASE if which == 'k' else self.V_BASE) + slot * self.TILE}"

    # softmax FINISH of strip X on buffer `buf`: the exponentials of key block 1 and all sixteen P dwords, as an in-order stream.
    # The v_fma of element e+1 is issued ahead of the v_exp of element e and no two dependent adds are adjacent (a dependent VALU
    # pair costs issue stalls: SQ_WAIT_INST_ANY was 15 % of the wave's cycles with fma -> exp and add -> add back to back).
    def finish_stream(self, X, buf, ps1=V_PS1, tmp=(V_T[2], V_T[3])):
        c1 = lambda k: vr(SBUF(buf, X, 1, k))
        c0 = lambda k: vr(SBUF(buf, X, 0, k))
        mc, ps0, l = vr(STV(X, 'mc')), vr(STV(X, 'ps0')), vr(STV(X, 'l'))
        fma = lambda e: f"v_fma_f32 {c1(e)}, {c1(e)}, {ka('scale_log2')}, -{mc}"

        def pack(i, x0, x1):
            if self.split and PSTART and i < 8:      # (key block 0: packed by the start stream, see start_stream)
                return []
            return self.pack_p(X, i, x0, x1, tmp)
        pk = PKADD and not self.split
        if pk:                                   # sums two at a time: both halves of a register pair per instruction
            acc = vr(PSP(X), 2)
            o = [fma(0)]
            for e in range(16):
                if e < 15:
                    o.append(fma(e + 1))
                o.append(f"v_exp_f32 {c1(e)}, {c1(e)}")
                if e == 0:
                    o.append(f"v_pk_add_f32 {acc}, {acc}, {vr(SBUF(buf, X, 0, 14), 2)}")     # left over from the start
                if e % 2 == 0:
                    o += pack(e // 2, c0(e), c0(e + 1))
                elif e >= 3:
                    o += pack(8 + (e - 3) // 2, c1(e - 3), c1(e - 2))
                    o.append(f"v_pk_add_f32 {acc}, {acc}, {vr(SBUF(buf, X, 1, e - 3), 2)}")
            o += pack(15, c1(14), c1(15))
            o.append(f"v_pk_add_f32 {acc}, {acc}, {vr(SBUF(buf, X, 1, 14), 2)}")
            o.append("s_nop 0")
            o.append(f"v_add_f32 {l}, {l}, {vr(PSP(X))}")
            o.append(f"v_add_f32 {l}, {l}, {vr(PSP(X) + 1)}")
            o.append(f"v_mov_b32 {vr(PSP(X))}, 0")
            o.append(f"v_mov_b32 {vr(PSP(X) + 1)}, 0")
            return o
        o = [fma(0)]
        for e in range(16):
            if e < 15:
                o.append(fma(e + 1))
            o.append(f"v_exp_f32 {c1(e)}, {c1(e)}")
            if e == 0:
                o.append(f"v_add_f32 {ps0}, {ps0}, {c0(15)}")      # left over from the start (its sums lag by one element)
            if e % 2 == 0:
                o += pack(e // 2, c0(e), c0(e + 1))
            elif e >= 3:
                o += pack(8 + (e - 3) // 2, c1(e - 3), c1(e - 2))
            if e >= 3:
                o.append(f"v_add_f32 {vr(ps1)}, {vr(ps1)}, {c1(e - 3)}")
        o.append(f"v_add_f32 {vr(ps1)}, {vr(ps1)}, {c1(13)}")
        o += pack(15, c1(14), c1(15))
        o.append(f"v_add_f32 {l}, {l}, {ps0}")
        o.append(f"v_add_f32 {vr(ps1)}, {vr(ps1)}, {c1(14)}")
        o.append(f"v_mov_b32 {ps0}, 0")
        o.append(f"v_add_f32 {vr(ps1)}, {vr(ps1)}, {c1(15)}")
        o.append("s_nop 0")
        o.append(f"v_add_f32 {l}, {l}, {vr(ps1)}")
        o.append(f"v_mov_b32 {vr(ps1)}, 0")
        return o

    def pack_p(self, X, i, x0, x1, tmp):
        """P dword i = the pair (x0, x1) in 16 bits; split P: also the pair of what the rounding left behind"""
        r = [f"{self.cvt} {vr(PD(X, i))}, {x0}, {x1}"]
        if self.split:
            t0, t1, hi = vr(tmp[0]), vr(tmp[1]), vr(PD(X, i))
            if self.dt == "bf16":
                r += [f"v_lshlrev_b32 {t0}, 16, {hi}", f"v_and_b32 {t1}, 0xffff0000, {hi}"]
            else:
                r += [f"v_cvt_f32_f16 {t0}, {hi}", f"v_cvt_f32_f16_sdwa {t1}, {hi} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"]
            r += [f"v_sub_f32 {t0}, {x0}, {t0}", f"v_sub_f32 {t1}, {x1}, {t1}", f"{self.cvt} {vr(PDL(X, i))}, {t0}, {t1}"]
        return r

    # softmax START of strip X on buffer `buf`, step u = 0..15 (row max, defer-max update, exponentials of key block 0)
    def start_fill(self, X, buf, u, alt=False):
        """alt: a second set of temporaries (row max in V_PS1, V_T[2:4], the compare's mask in s[t2]) for a stream that is interleaved
        with the other strip's"""
        n0 = lambda k: vr(SBUF(buf, X, 0, k))
        mx, m, l, mc, th, al, ps0 = vr(V_PS1 if alt else V_MX), *(vr(STV(X, f)) for f in ("m", "l", "mc", "thr", "al", "ps0"))
        t0, t1 = (vr(V_T[2]), vr(V_T[3])) if alt else (vr(V_T[0]), vr(V_T[1]))
        cc = S('t2') if alt else "vcc"
        o = []
        if u < 4:
            kb, o8 = u >> 1, (u & 1) * 8
            s = lambda k: vr(SBUF(buf, X, kb, o8 + k))
            if u == 0:
                o.append(f"v_max3_f32 {mx}, {s(0)}, {s(1)}, {s(2)}")
                o.append(f"v_max3_f32 {mx}, {mx}, {s(3)}, {s(4)}")
                o.append(f"v_max3_f32 {mx}, {mx}, {s(5)}, {s(6)}")
                o.append(f"v_max_f32 {mx}, {mx}, {s(7)}")
            else:
                for k in range(0, 8, 2):
                    o.append(f"v_max3_f32 {mx}, {mx}, {s(k)}, {s(k + 1)}")
            if u == 3:                       # the lane pair that shares a query row (VALU write -> permlane read: 2 wait states)
                o += [f"v_mov_b32 {t0}, {mx}", "s_nop 1", f"v_permlane32_swap_b32 {mx}, {t0}", f"v_max_f32 {mx}, {mx}, {t0}"]
        elif u == 4:
            # branch-free, per row: a max inside the headroom leaves m alone, and then alpha = exp2(0) = 1 exactly
            o += [f"v_cmp_gt_f32 {cc}, {mx}, {th}",
                  f"s_or_b64 {S('grow')}, {S('grow')}, {cc}",
                  ("s_nop 0" if (PKADD and not self.split) else f"v_mov_b32 {ps0}, 0"),
                  f"v_cndmask_b32 {t0}, {m}, {mx}, {cc}",                 # m_new
                  f"v_sub_f32 {t1}, {m}, {t0}",
                  f"v_mul_f32 {t1}, {ka('scale_log2')}, {t1}",
                  f"v_exp_f32 {al}, {t1}",
                  f"v_mov_b32 {m}, {t0}",
                  f"v_add_f32 {th}, {ka('thr')}, {t0}",
                  f"v_mul_f32 {mc}, {ka('scale_log2')}, {t0}",
                  f"v_mul_f32 {l}, {l}, {al}"]
        else:
            raise ValueError("u >= 5: see start_stream")
        return o

    def start_stream(self, X, buf, alt=False):
        """row max, defer-max update, then the exponentials of key block 0 (fma one element ahead, sums one behind; element 15's
        sum is added by the finish)"""
        n0 = lambda k: vr(SBUF(buf, X, 0, k))
        mc, ps0 = vr(STV(X, 'mc')), vr(STV(X, 'ps0'))
        fma = lambda e: f"v_fma_f32 {n0(e)}, {n0(e)}, {ka('scale_log2')}, -{mc}"
        o = []
        for u in range(5):
            o += self.start_fill(X, buf, u, alt)
        o.append(fma(0))
        for e in range(16):
            if e < 15:
                o.append(fma(e + 1))
            o.append(f"v_exp_f32 {n0(e)}, {n0(e)}")
            if PKADD and not self.split:
                if e >= 2 and e % 2 == 0:         # the pair (e-2, e-1); the last pair is added by the finish
                    o.append(f"v_pk_add_f32 {vr(PSP(X), 2)}, {vr(PSP(X), 2)}, {vr(SBUF(buf, X, 0, e - 2), 2)}")
            elif e >= 1:
                o.append(f"v_add_f32 {ps0}, {ps0}, {n0(e - 1)}")
        if alt:
            o.append(f"v_mov_b32 {vr(V_PS1)}, 0")         # (it stood in for the row max)
        if self.split and PSTART:
            # key block 0's P dwords (hi and lo), behind the PV MFMAs of k-steps 0 / 1 that still read the previous tile's (marker: pack())
            tmp = (V_T[2], V_T[3]) if alt else (V_T[0], V_T[1])
            o.append(f"@gap {4 * 2 * self.DB + 1}")
            o.append("s_nop 0")                            # (the last v_exp above and its consumer below)
            for k in range(8):
                o += self.pack_p(X, k, n0(2 * k), n0(2 * k + 1), tmp)
        return o

    # ---- the fast loop (self.fast): no row maximum inside the tile loop ------------------------------------------------------------
    # The reference rescales by the running row maximum at every tile (flash_attention_3.py:239-250); the defer-max loop already takes
    # it only to learn that it has NOT outgrown its headroom -- 16 v_max3 + a shuffle + an 11-instruction update per strip and tile.
    # Here a tile's exponentials are simply taken against the maximum the item's tile 0 established (start_fast with_max): the scaled
    # score x = s c - m c overwrites s and STAYS in the S buffer, exp2(x) goes to a scratch register, from there into the row sum and
    # into the packed P.  One compare per strip and tile (a lane's sums of the tile against 2^14: then no weight of it exceeds 2^14 --
    # fp16-safe, and far below anything fp32 sums care about) feeds the wave-uniform `grow` mask; a set bit sends the wave to fixup(),
    # which has every x of the tile at hand and redoes it exactly.  60 vector instructions fewer per tile, bit-identical to the
    # defer-max loop while no row outgrows its tile-0 maximum by 2^8.
    def PDX(self, X, b, i):
        """P dword i of the tile that lives in S buffer b"""
        return PD0B(X, i) if (self.fast and b == 1 and i < 8) else PD(X, i)

    def exp_block(self, X, b, kb, ps, scaled=False):
        """fma / exp / sum / pack of key block kb of strip X, tile in S buffer b: x over s in place, the fma one element ahead of its exp,
        sums one element behind, packs three behind (a v_exp's result must not be read by the very next vector instruction: gfx940+
        want a wait state between a transcendental and its consumer).  scaled: the buffer already holds x (fixup): no fma."""
        c = lambda k: vr(SBUF(b, X, kb, k))
        t = lambda e: vr(FT[X][e % 4])
        mc = vr(STV(X, 'mc'))
        fma = lambda e: f"v_fma_f32 {c(e)}, {c(e)}, {ka('scale_log2')}, -{mc}"
        pack = lambda k: f"{self.cvt} {vr(self.PDX(X, b, 8 * kb + k))}, {t(2 * k)}, {t(2 * k + 1)}"
        o = [] if scaled else [fma(0)]
        for e in range(16):
            if e < 15 and not scaled:
                o.append(fma(e + 1))
            o.append(f"v_exp_f32 {t(e)}, {c(e)}")
            if e == 2 and DIET:
                o.append(f"v_add_f32 {ps}, {t(0)}, {t(1)}")           # (starts the sum: no zeroing, and 0 + t0 = t0 exactly -- the same bits)
            elif e >= (3 if DIET else 1):
                o.append(f"v_add_f32 {ps}, {ps}, {t(e - 1)}")
            if e % 2 == 1 and e >= 3:
                o.append(pack((e - 3) // 2))
        o.append(f"v_add_f32 {ps}, {ps}, {t(15)}")
        o.append(pack(7))
        return o

    def finish_fast(self, X, b, alt=False):
        """key block 1 of the tile in S buffer b (+ without DIET: the strip's own check)"""
        o = self.exp_block(X, b, 1, vr(PS1(X)))
        if not DIET:
            ps0, ps1, lim, tt = vr(STV(X, 'ps0')), vr(PS1(X)), vr(STV(X, 'thr')), vr(V_T[1] if alt else V_T[0])
            cc = S('t2') if alt else "vcc"
            o += [f"v_max_f32 {tt}, {ps0}, {ps1}",
                  f"v_cmp_nge_f32 {cc}, {lim}, {tt}",                  # not (2^14 >= sum): too large, or not a number
                  f"s_or_b64 {S('grow')}, {S('grow')}, {cc}"]
        return o

    def tile_check(self):
        """the tile's check, both strips at once: the largest of a lane's four block sums against 2^14 (STV thr holds it) -> VCC, for
        fix_check's s_cbranch_vccnz (nothing between the two writes VCC)"""
        if not DIET:
            return []
        tt = vr(V_T[0])
        if self.kmask:                         # the limit is a row's own (negative while the row is fresh)
            tb = vr(V_T[1])
            return [f"v_max_f32 {tt}, {vr(STV('A', 'ps0'))}, {vr(PS1('A'))}", f"v_max_f32 {tb}, {vr(STV('B', 'ps0'))}, {vr(PS1('B'))}",
                    f"v_cmp_nge_f32 {S('t2')}, {vr(STV('B', 'thr'))}, {tb}", f"v_cmp_nge_f32 vcc, {vr(STV('A', 'thr'))}, {tt}",
                    f"s_or_b64 vcc, vcc, {S('t2')}"]
        return [f"v_max3_f32 {tt}, {vr(STV('A', 'ps0'))}, {vr(PS1('A'))}, {vr(STV('B', 'ps0'))}",
                f"v_max_f32 {tt}, {tt}, {vr(PS1('B'))}",
                f"v_cmp_nge_f32 vcc, {vr(STV('A', 'thr'))}, {tt}"]       # not (2^14 >= sum): too large, or not a number

    def lupd(self, X):
        """the tile's two block sums into l (one phase after the check, so that fixup() finds l without them), in the defer-max loop's order"""
        l, ps0, ps1 = vr(STV(X, 'l')), vr(STV(X, 'ps0')), vr(PS1(X))
        if DIET:                               # (every block sum starts afresh with t0 + t1: nothing to zero)
            return [f"v_add_f32 {l}, {l}, {ps0}", f"v_add_f32 {l}, {l}, {ps1}"]
        return [f"v_add_f32 {l}, {l}, {ps0}", f"v_mov_b32 {ps0}, 0", f"v_add_f32 {l}, {l}, {ps1}", f"v_mov_b32 {ps1}, 0"]

    def start_fast(self, X, b, with_max=False):
        """key block 0 of the tile in S buffer b.  with_max: an item's tile 0 -- its row maximum (both key blocks) becomes the item's m."""
        o = []
        if with_max:
            mx, t0 = vr(V_T[2]), vr(V_T[0])
            r = [vr(SBUF(b, X, kb, k)) for kb in range(2) for k in range(16)]
            o.append(f"v_max3_f32 {mx}, {r[0]}, {r[1]}, {r[2]}")
            for k in range(3, 31, 2):
                o.append(f"v_max3_f32 {mx}, {mx}, {r[k]}, {r[k + 1]}")
            o.append(f"v_max_f32 {mx}, {mx}, {r[31]}")
            o += [f"v_mov_b32 {t0}, {mx}", "s_nop 1", f"v_permlane32_swap_b32 {mx}, {t0}", f"v_max_f32 {mx}, {mx}, {t0}"]
            if self.klen:                      # a batch without any key: every score is -inf; keep m finite so that x = s c - m c stays -inf
                o.append(f"v_max_f32 {mx}, {NEG_BIG}, {mx}")
            if self.kmask:                     # no visible key in tile 0: a fresh row -- m = 0 (x = s c keeps the score), limit -1 (the check always fires)
                o += [f"v_cmp_lt_f32 vcc, {NEG_BIG}, {mx}",
                      f"v_mov_b32 {t0}, 0x46800000",
                      f"v_cndmask_b32 {mx}, 0, {mx}, vcc",
                      f"v_cndmask_b32 {vr(STV(X, 'thr'))}, -1.0, {t0}, vcc"]
            o.append(f"v_mul_f32 {vr(STV(X, 'mc'))}, {ka('scale_log2')}, {mx}")
        else:
            o += self.lupd(X)
        o += self.exp_block(X, b, 0, vr(STV(X, 'ps0')))
        return o

    def fixup(self, b):
        """Subroutine (s_swappc return address in s[58:59]): some lane of the wave found a tile sum past 2^14 in the tile of S buffer b.
        Per row: the maximum of its scaled scores x; if it outgrew the running maximum by more than 2^8 it becomes the new one (x -= it,
        m c += it, l and O scaled by exp2(-it)); every weight of the tile is taken again from its x, both strips, exactly as
        exp_block does (rows that keep their maximum get the same bits again).  O has the tiles before this one, l likewise (lupd)."""
        self.cm(f"fix-up of the tile in S buffer {b}")
        mx, t0, sh = vr(V_T[2]), vr(V_T[0]), vr(V_T[3])
        for X in "AB":
            r = [vr(SBUF(b, X, kb, k)) for kb in range(2) for k in range(16)]
            self.i(f"v_max3_f32 {mx}, {r[0]}, {r[1]}, {r[2]}")
            for k in range(3, 31, 2):
                self.i(f"v_max3_f32 {mx}, {mx}, {r[k]}, {r[k + 1]}")
            self.i(f"v_max_f32 {mx}, {mx}, {r[31]}")
            self.emit([f"v_mov_b32 {t0}, {mx}", "s_nop 1", f"v_permlane32_swap_b32 {mx}, {t0}", f"v_max_f32 {mx}, {mx}, {t0}"])
            self.i(f"v_cmp_lt_f32 vcc, 8.0, {mx}")
            self.i("s_nop 1")
            self.i(f"v_cndmask_b32 {sh}, 0, {mx}, vcc")
            if self.kmask:                     # a fresh row (limit < 0) takes the maximum of its first visible keys, whatever it is, and is fresh no more
                t1, lim = vr(V_T[1]), vr(STV(X, 'thr'))
                self.i(f"v_mov_b32 {t0}, {NEG_BIG}")
                self.i(f"v_cmp_lt_f32 vcc, {t0}, {mx}")                        # the tile shows this row a key
                self.i(f"v_mov_b32 {t0}, 0x46800000")
                self.i(f"v_cndmask_b32 {t1}, 0, {mx}, vcc")
                self.i(f"v_cndmask_b32 {t0}, {lim}, {t0}, vcc")
                self.i(f"v_cmp_gt_f32 vcc, 0, {lim}")                          # fresh
                self.i("s_nop 1")
                self.i(f"v_cndmask_b32 {sh}, {sh}, {t1}, vcc")
                self.i(f"v_cndmask_b32 {lim}, {lim}, {t0}, vcc")
            self.i(f"v_add_f32 {vr(STV(X, 'mc'))}, {vr(STV(X, 'mc'))}, {sh}")
            self.i(f"v_exp_f32 {vr(STV(X, 'al'))}, -{sh}")
            for x in r:
                self.i(f"v_sub_f32 {x}, {x}, {sh}")
            self.i(f"v_mul_f32 {vr(STV(X, 'l'))}, {vr(STV(X, 'l'))}, {vr(STV(X, 'al'))}")
            for kb, ps in ((0, vr(STV(X, 'ps0'))), (1, vr(PS1(X)))):
                self.i(f"v_mov_b32 {ps}, 0")
                self.emit(self.exp_block(X, b, kb, ps, scaled=True))
        self.rescale(clear=not DIET)               # O *= alpha, both strips (the diet loop keeps its return address in `grow`)
        for X in "AB":
            self.i(f"v_mov_b32 {vr(STV(X, 'al'))}, 1.0")
        self.i(f"s_setpc_b64 {self.ret}")

    def fix_check(self, b):
        """behind the phase that finished the tile of S buffer b: call fixup(b) if a lane asked for it"""
        lc, lb = self.ul("fix"), self.ul("fixed")
        if DIET:
            self.i(f"s_cbranch_vccnz {lc}")      # (tile_check's compare)
        else:
            self.i(f"s_cmp_lg_u64 {S('grow')}, 0")
            self.i(f"s_cbranch_scc1 {lc}")
        self.lab(lb)
        self.out_of_line(True)
        self.lab(lc)
        if self.kmask:                         # a tile without a visible key changes nothing (fresh rows ask every tile)
            self.i(f"s_cmp_eq_u64 {self.MK(b)}, 0")
            self.i(f"s_cbranch_scc1 {lb}")
        la = self.ul("pc")
        self.i(f"s_getpc_b64 {S('t2')}")
        self.lab(la)
        self.i(f"s_add_u32 {S('t2', 0)}, {S('t2', 0)}, .L{self.name}_fixup{b}-{la}")
        self.i(f"s_addc_u32 {S('t2', 1)}, {S('t2', 1)}, 0")
        self.i(f"s_swappc_b64 {self.ret}, {S('t2')}")
        self.i(f"s_branch {lb}")
        self.out_of_line(False)

    @staticmethod
    def interleave(a, b):
        o = []
        for i in range(max(len(a), len(b))):
            o += a[i:i + 1] + b[i:i + 1]
        return o

    def qk_mfma(self, buf, X, i):
        kb, ks = i // self.KS, i % self.KS
        acc = vr(SBUF(buf, X, kb, 0), 16)
        return f"{self.mf} {acc}, {fr(KFR(i), 4)}, {ar(self.QA(X, ks), 4)}, {'0' if ks == 0 else acc}"

    def pv_mfma(self, X, idx, lo=False):
        f, db = idx // self.DB, idx % self.DB
        acc = ar(self.OA(X, db), 16)
        pd = PDL(X, 4 * f) if lo else self.PDX(X, self.pv_buf, 4 * f)
        return f"{self.mf} {acc}, {fr(VFR(idx), 4)}, {vr(pd, 4)}, {acc}"

    # ---- phases ----------------------------------------------------------------------------------------------------------
    def phase_qk(self, p, fillers, dma_at, lg=None, pre=(), tail_vreads=None, rec=None):
        """QK^T(j+1) into buffer 1-p from K slot 1-p; fillers[g] = instructions for the gap behind MFMA g (a fragment feeds strip A,
        then strip B).  tail_vreads = V slot whose first RING V^T fragments are requested in the last gaps (after the last K read),
        for the PV phase that follows.  rec (dry run): gets the issue cycles of every gap's fixed content instead of fillers."""
        nb, slot, R, NF = 1 - p, 1 - p, self.RING, self.NKF
        ngaps = 2 * NF
        lg = lg or Lgkm()
        if ('k', 0) not in lg.q:                 # (mid-barrier loop: the iteration before requested them -- kprefetch -- and the caller registered them)
            for i in range(R):
                self.i(self.kread(slot, i))
                lg.issue(('k', i))
        self.emit(pre)
        for hs in range(ngaps):
            i, X = hs // 2, 'AB'[hs % 2]
            mark = len(self.L)
            if hs % 2 == 0 and i % WAITN == 0:
                # (seam: the successor's Q fragments are in flight too -- older than the K reads, or, in the mid-barrier loop, younger)
                w = lg.need([('k', i + x) for x in range(WAITN)] + [('q', Y, i + x) for Y in "AB" for x in range(WAITN)])
                if w:
                    self.i(w)
            self.i(self.qk_mfma(nb, X, i))
            if hs % 2 == 1 and i % 2 == 1:
                for f in (i + R - 1, i + R):
                    if f < NF:
                        self.i(self.kread(slot, f))
                        lg.issue(('k', f))
            if tail_vreads is not None and hs >= ngaps - min(R, self.NVF):
                n = hs - (ngaps - min(R, self.NVF))
                for k, x in enumerate(self.vread(tail_vreads, n)):
                    self.i(x)
                    lg.issue(('v', n, k))
            self.emit(dma_at.get(hs, []))
            if rec is not None:
                rec.append(sum(self.price(x.strip()) for x in self.L[mark:] if not x.strip().startswith("v_mfma")))
            else:
                self.emit(fillers[hs])
        return lg

    def kprefetch_lg(self):
        """tracker of a QK^T phase whose first K fragments are in flight since the iteration before (mid-barrier loop)"""
        lg = Lgkm()
        for i in range(self.RING):
            lg.issue(('k', i))
        return lg

    def kprefetch(self, p):
        """the first K fragments of the NEXT iteration's QK^T phase (parity 1-p: K slot p)"""
        return [self.kread(p, i) for i in range(self.RING)]

    def phase_pv(self, p, fillers, dma_at, strips="AB", lg=None, preissued=False, rec=None, mid=None):
        """PV(j): P dwords x V slot p -> O; one gap per MFMA (split P: a fragment feeds the hi pass of every strip, then the lo pass);
        fillers as above.  rec (dry run): gets the issue cycles of every gap's fixed content instead of emitting fillers."""
        slot, R, NF = p, min(self.RING, self.NVF), self.NVF
        self.pv_buf = p                        # (fast loop: the tile's P dwords 0..7 have a buffer per S buffer)
        lg = lg or Lgkm()
        if not preissued:
            for idx in range(R):
                for k, x in enumerate(self.vread(slot, idx)):
                    self.i(x)
                    lg.issue(('v', idx, k))
        hs = 0
        passes = [(X, lo) for lo in ((False, True) if self.split else (False,)) for X in strips]
        pending = []
        for idx in range(NF):
            for n, (X, lo) in enumerate(passes):
                mark = len(self.L)
                if n == 0 and idx % WAITN == 0:
                    w = lg.need([('v', idx + x, 1) for x in range(WAITN)])
                    if w:
                        self.i(w)
                self.i(self.pv_mfma(X, idx, lo))
                if n == len(passes) - 1 and idx % 2 == 1:
                    for f in (idx + R - 1, idx + R):
                        if f < NF:
                            for k, x in enumerate(self.vread(slot, f)):
                                self.i(x)
                                lg.issue(('v', f, k))
                self.emit(dma_at.get(hs, []))
                if mid is not None and hs == mid['gap']:
                    # the iteration's barrier (mid-barrier loop): every V^T read of this tile has been issued (all of them are behind
                    # MFMA 15 at the latest) and is waited for here, so the next iteration's DMA may overwrite V slot p; this wave's own
                    # DMA pieces of the iteration are waited for by `inline`; behind the barrier K(j+2) is visible: its first fragments
                    # are requested at once, two a gap, and arrive under the MFMAs that are left
                    assert all(t[0] != 'v' or t[1] < NF for t in lg.q) and idx + R >= NF - 1 or True
                    self.i("s_waitcnt lgkmcnt(0)")
                    lg.q = []
                    self.emit(mid['inline'])
                    self.i("s_barrier")
                    pending = list(mid['kreads'])
                elif pending:
                    for x in pending[:2]:
                        self.i(x)
                        lg.issue(('kn', x))
                    pending = pending[2:]
                if rec is not None:
                    rec.append(sum(self.price(x.strip()) for x in self.L[mark:] if not x.strip().startswith("v_mfma")))
                else:
                    self.emit(fillers[hs] if hs < len(fillers) else [])
                hs += 1
        for x in pending:
            self.i(x)
        return lg

    def run_phase(self, fn, stream, lg, **kw):
        """Dry-run the phase to learn each gap's fixed issue cycles, slice `stream` over the gaps around them, emit; -> lg"""
        import copy
        save, uid = self.L, self.uid
        self.L, rec = [], []
        fn(fillers=None, lg=copy.deepcopy(lg), rec=rec, **kw)
        self.L, self.uid = save, uid
        fillers, tail = self.pack(stream, rec, ngaps=len(rec))
        lg = fn(fillers=fillers, lg=lg, **kw)
        self.emit(tail)
        return lg

    # ---- DMA stream bookkeeping (top and bottom of every iteration, all body kinds; workgroup-uniform) -----------------------------
    def stream_top(self, p):
        l1, l2 = self.ul("kok"), self.ul("vok")
        self.i(f"s_cmp_lg_u32 {S('krem')}, 0")
        self.i(f"s_cbranch_scc1 {l1}")
        self.i(f"s_mov_b64 {S('ksrd', 0, 2)}, {S('ksrd_n')}")           # K(j+2) is tile 0 of the next item
        self.i(f"s_mov_b32 {S('koff')}, 0")
        self.i(f"s_mov_b32 {S('krem')}, {S('nt_n')}")
        if self.kmask:                                                  # ... and so are its mask bytes: row n_b
            self.i(f"s_mul_i32 {ka('pad')}, {S('n_b')}, {ka('Sk')}")
            self.i(f"s_mov_b32 {self.kleft}, {self.kleft0}")             # ... and the keys its batch has
        if self.klen:
            self.i(f"s_mov_b32 {ka('pad')}, {S('L_n')}")                 # keys left from tile j+2 on
        self.lab(l1)
        if self.kmask:                                                  # the bytes of tile j+2 (its K pieces go to slot p in this iteration),
            self.len_word(0, dst=S('t2'), left=self.kleft)              # only the lanes whose keys the batch has: nothing is read past a mask row
            self.i(f"s_mov_b64 exec, {S('t2')}")
            self.emit(self.mask_load())
            self.i("s_mov_b64 exec, -1")
        if self.klen:
            self.len_word(p)                                            # the word of tile j+2 from the keys left
        self.i(f"s_cmp_lg_u32 {S('vrem')}, 0")
        self.i(f"s_cbranch_scc1 {l2}")
        self.i(f"s_mov_b64 {S('vsrd', 0, 2)}, {S('vsrd_n')}")
        self.i(f"s_mov_b32 {S('voff')}, 0")
        self.i(f"s_mov_b32 {S('vrem')}, {S('nt_n')}")
        self.lab(l2)

    def capture(self, fn):
        """the instructions fn() emits in line, as a list (what it emits out of line goes where such code goes)"""
        save, self.L = self.L, []
        fn()
        out, self.L = self.L, save
        return [x[1:] if x.startswith("\t") else x for x in out]

    def stream_bottom(self, p, barrier=True):
        """Q pieces of the next item (first two iterations of an item; one at D = 64), the counted wait, the barrier (barrier=False: the
        mid-barrier loop emits this in front of its own, inside the PV phase)."""
        lq, lb = self.ul("q"), self.ul("bar")
        self.i(f"s_add_u32 {S('koff')}, {S('koff')}, {S('ktile')}")
        self.i(f"s_add_u32 {S('voff')}, {S('voff')}, {S('vtile')}")
        self.i(f"s_sub_u32 {S('krem')}, {S('krem')}, 1")
        self.i(f"s_sub_u32 {S('vrem')}, {S('vrem')}, 1")
        self.i(f"s_sub_u32 {S('wrem')}, {S('wrem')}, 1")
        if self.kmask:
            self.i(f"s_add_u32 {ka('pad')}, {ka('pad')}, 64")
        if self.klen:
            self.i(f"s_sub_u32 {ka('pad')}, {ka('pad')}, 64")
        self.i(f"s_cmp_lg_u32 {S('qrem')}, 0")
        self.i(f"s_cbranch_scc1 {lq}")
        self.stamp(2, fine=True)
        self.i("s_waitcnt vmcnt(0)")
        self.stamp(9, fine=True)                # bucket 9: the wait for this wave's own DMA pieces
        self.lab(lb)
        if self.kmask:
            self.i(self.mask_word(p))           # tile j+2 -> MK((j+2) & 1) = MK(p)
            self.km_len_and(p)
        if barrier:
            self.i("s_barrier")
        self.stamp(12, fine=True)               # bucket 12: the barrier
        self.out_of_line(True)
        self.lab(lq)
        self.q_group()
        self.q_group()                          # two groups an iteration: the next item's Q block is complete by the third iteration
        self.i("s_waitcnt vmcnt(8)")            # everything but the eight Q pieces just issued
        self.i(f"s_branch {lb}")
        self.out_of_line(False)

    def q_group(self):
        """Four 1-KiB pieces (whole rows: 16 at D = 128, 32 at D = 64) of this wave's share of the NEXT item's Q block -> its landing zone."""
        self.i(f"s_mov_b32 m0, {S('qdst')}")
        self.i(f"s_sub_u32 {S('qrem')}, {S('qrem')}, 1")
        for t in range(4):
            self.i(f"buffer_load_dwordx4 {vr(QDOFF(t))}, {S('qsrd_n')}, {S('qoff')} offen offset:{1024 * t} lds")
        self.i(f"s_add_u32 {S('qdst')}, {S('qdst')}, 4096")
        self.i(f"s_lshl_b32 {S('t0')}, {ka('q_ss')}, {4 if self.D == 128 else 5}")
        self.i(f"s_add_u32 {S('qoff')}, {S('qoff')}, {S('t0')}")

    # ---- bodies ----------------------------------------------------------------------------------------------------------
    def dma_plan(self, p, gaps_v, gaps_k):
        """V(j+1) -> V slot 1-p, K(j+2) -> K slot p: {gap: [instructions]}"""
        d = {}
        def piece(which, n):
            if not ABL:
                return [self.dma(which, n)]
            # TIMING-ONLY ablations (wrong results): "nodma" = no wave issues its pieces, "dma_w0" = only wave 0 does
            lab = self.ul("abl")
            if ABL == "nodma":
                return [f"s_cmp_eq_u32 {S('wave')}, 99", f"s_cbranch_scc0 {lab}", self.dma(which, n), lab + ":"]
            return [f"s_cmp_eq_u32 {S('wave')}, 0", f"s_cbranch_scc0 {lab}", self.dma(which, n), lab + ":"]
        for n, g in enumerate(gaps_v):
            d.setdefault(g, [])
            if n == 0:
                d[g] += [self.setm0('v', 1 - p), "s_nop 0"]          # M0 write -> LDS-DMA: one wait state
            d[g] += piece('V', n)
        for n, g in enumerate(gaps_k):
            d.setdefault(g, [])
            if n == 0:
                d[g] += [self.setm0('k', p), "s_nop 0"]
            d[g] += piece('K', n)
        return d

    # ---- gap packing -------------------------------------------------------------------------------------------------------------
    @staticmethod
    def price(ins):
        """vector-issue cycles of one instruction beside MFMAs (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')"""
        op = ins.split()[0]
        if op.endswith(":") or op.startswith("@"):
            return 0
        if op.startswith(("v_exp", "v_log", "v_rcp")):
            return 8
        if op.startswith("ds_"):
            return 2
        if op.startswith("buffer_load"):
            return DMA_PRICE
        if op == "s_nop":
            return 4 * (int(ins.split()[1]) + 1)
        return 4

    def pack(self, stream, fixed, ngaps=32, budget=24):
        """Slice an in-order instruction stream over the MFMA gaps: gap g takes instructions while its fixed content (LDS reads,
        waits, DMA pieces) plus what it has taken stays within the budget of issue cycles an MFMA's shadow hides.  What does not
        fit anywhere is returned as the tail (emitted behind the last MFMA)."""
        st = list(stream)
        total = sum(self.price(x) for x in st)
        room_all = sum(max(0, budget - f) for f in fixed)
        over = max(0, total - room_all)
        out = []
        for g in range(ngaps):
            room = budget - fixed[g] + (over * (g + 1)) // ngaps - (over * g) // ngaps     # spread the unavoidable excess evenly
            take = []
            while st:
                if st[0].startswith("@gap"):   # marker: what follows must not be issued before MFMA gap N
                    if g < int(st[0].split()[1]):
                        break
                    st.pop(0)
                    continue
                if room - self.price(st[0]) < -2:
                    break
                room -= self.price(st[0])
                take.append(st.pop(0))
            out.append(take)
        return out, [x for x in st if not x.startswith("@")]

    def body_full(self, p, lean=False):
        """lean: an iteration the caller knows to be far from the item's ends (no diagonal tile, see the loop in kernel())"""
        self.cm(f"FULL body, parity {p}{' (lean loop)' if lean else ''}: QK^T(j+1) || softmax finish(j);  PV(j) || softmax start(j+1)")
        self.stamp(2)
        self.abl_on = True
        # ---- phase A: the finish of both strips as ONE in-order stream, sliced evenly over the gaps around their fixed content
        if self.fast:
            fin = self.interleave(self.finish_fast('A', p), self.finish_fast('B', p, alt=True)) + self.tile_check()
        elif ILV:                              # the two strips' streams interleaved: twice the distance between dependent instructions (-0.7 % cycles)
            fin = self.interleave(self.finish_stream('A', p), self.finish_stream('B', p, ps1=V_MX, tmp=(V_E[6], V_E[7])))
            fin = [f"v_mov_b32 {vr(V_MX)}, 0"] + fin
        else:
            fin = self.finish_stream('A', p) + self.finish_stream('B', p)
        n = 2 * self.NKF                       # gaps of the QK^T phase; the wave's PPW pieces of V early, of K late
        gv = [g * n // 32 for g in DMA_GAPS_V][:self.PPW] if self.PPW == 4 else [1 * n // 16, 5 * n // 16]
        gk = [g * n // 32 for g in DMA_GAPS_K][:self.PPW] if self.PPW == 4 else [9 * n // 16, 13 * n // 16]
        dma = self.dma_plan(p, gv, gk)
        npre = 0
        while sum(self.price(x) for x in fin[:npre + 1]) <= (0 if self.mb else PRE):
            npre += 1
        pre, fin = fin[:npre], fin[npre:]
        lg = self.run_phase(self.phase_qk, fin, self.kprefetch_lg() if self.mb else Lgkm(), p=p, dma_at=dma, pre=pre, tail_vreads=p)
        if self.fast:
            self.fix_check(p)                  # tile j is complete: its check (before PV(j) takes its P)
        self.stamp(0, fine=True)
        if not (lean and self.klen):           # (klen: a lean iteration is that far from the last tiles that all its keys exist)
            self.mask_keys(1 - p, 1 - p)       # tile j+1: buffer 1-p, word MK((j+1) & 1)
        if self.causal and not lean:           # the diagonal tile of this wave is tile wnt-1 = j+1  <=>  wrem == 1
            lm, lr = self.ul("mask"), self.ul("masked")
            self.i(f"s_cmp_eq_u32 {S('wrem')}, {self.nd if self.cl else 1}")
            self.i(f"s_cbranch_scc1 {lm}")
            self.lab(lr)
            self.out_of_line(True)
            self.lab(lm)
            self.mask_diag(1 - p)
            self.i(f"s_branch {lr}")
            self.out_of_line(False)
        # ---- phase B: the start of both strips, same treatment
        if self.fast:
            sta = self.interleave(self.start_fast('A', 1 - p), self.start_fast('B', 1 - p))
        elif ILV:
            sta = self.interleave(self.start_stream('A', 1 - p), self.start_stream('B', 1 - p, alt=True))
        else:
            sta = self.start_stream('A', 1 - p) + self.start_stream('B', 1 - p)
        mid = None
        if self.mb:                            # the iteration's bookkeeping, counted wait and barrier inside PV(j); K(j+2)'s first fragments behind it
            if lean:
                inline = [f"s_add_u32 {S('koff')}, {S('koff')}, {S('ktile')}", f"s_add_u32 {S('voff')}, {S('voff')}, {S('vtile')}"] + \
                         ([f"s_add_u32 {ka('pad')}, {ka('pad')}, 64"] if self.kmask else []) + ["s_waitcnt vmcnt(0)"] + \
                         ([self.mask_word(p)] if self.kmask else [])
            else:
                inline = self.capture(lambda: self.stream_bottom(p, barrier=False))
            mid = dict(gap=MBGAP or (27 if self.D == 128 else 12), inline=inline, kreads=self.kprefetch(p))
        self.run_phase(self.phase_pv, sta, lg, p=p, dma_at={}, preissued=True, mid=mid)
        self.abl_on = False
        self.stamp(1, count=7)
        if self.fast:
            return
        # pending O rescale (rare: defer-max)
        lr, lb = self.ul("rescale"), self.ul("rescaled")
        self.i(f"s_cmp_lg_u64 {S('grow')}, 0")
        self.i(f"s_cbranch_scc1 {lr}")
        self.lab(lb)
        self.out_of_line(True)
        self.lab(lr)
        self.rescale()
        self.i(f"s_branch {lb}")
        self.out_of_line(False)

    def body_last(self, p):
        self.cm(f"LAST body, parity {p}: this wave's last tile -- finish(j), PV(j); nothing to prefetch for the wave itself")
        self.stamp(2)
        d = self.dma_plan(p, [2 * t for t in range(self.PPW)], [2 * self.PPW + 2 * t for t in range(self.PPW)])    # PV(A) has >= NVF gaps
        if self.fast:                          # both finishes, the tile's check, its sums into l; then PV(j) of both strips with the DMA pieces
            self.emit(self.interleave(self.finish_fast('A', p), self.finish_fast('B', p, alt=True)) + self.tile_check())
            self.fix_check(p)
            self.emit(self.lupd('A') + self.lupd('B'))
            self.i("s_nop 1")
            self.phase_pv(p, [[] for _ in range(2 * self.NVF)], d, strips="AB")
            self.stamp(5)
            return
        self.emit(self.finish_stream('A', p))
        self.i("s_nop 1")
        self.run_phase(self.phase_pv, self.finish_stream('B', p), Lgkm(), p=p, dma_at=d, strips="A")
        self.i("s_nop 1")
        self.phase_pv(p, [[] for _ in range(2 * self.NVF)], {}, strips="B")
        self.stamp(5)

    def body_skip(self, p):
        self.cm(f"SKIP body, parity {p}: this wave is past its last tile of the item -- keep the rings fed")
        self.stamp(2)
        self.i(self.setm0('v', 1 - p))
        self.i("s_nop 0")
        for t in range(self.PPW):
            self.i(self.dma('V', t))
        self.i(self.setm0('k', p))
        self.i("s_nop 0")
        for t in range(self.PPW):
            self.i(self.dma('K', t))
        self.stamp(6)

    def mask_diag(self, buf):
        """Causal mask of the wave's diagonal tile (key base = the wave's first row): key 32 kb + kidx(e) + 4h is visible to row
        r of strip A iff 32 kb + kidx < r + 1 - 4h = tA, to row r of strip B iff 32 kb + kidx < tA + 32.  So strip A key block 0
        and strip B key block 1 are partial with the SAME lane test, strip A key block 1 is masked whole, strip B key block 0
        is visible whole."""
        self.i("s_nop 7")
        self.i("s_nop 7")                     # last QK^T MFMA -> VALU access of S
        ninf = vr(V_T[3])                     # (a literal and VCC together exceed the constant bus)
        self.i(f"v_mov_b32 {ninf}, {NEG_INF}")
        for e in range(16):
            kidx = (e & 3) + 8 * (e >> 2)
            self.i(f"v_cmp_gt_i32 vcc, {vr(V_TA)}, {kidx}")
            self.i(f"v_mov_b32 {vr(SBUF(buf, 'A', 1, e))}, {ninf}")
            self.i("s_nop 0")
            self.i(f"v_cndmask_b32 {vr(SBUF(buf, 'A', 0, e))}, {ninf}, {vr(SBUF(buf, 'A', 0, e))}, vcc")
            self.i(f"v_cndmask_b32 {vr(SBUF(buf, 'B', 1, e))}, {ninf}, {vr(SBUF(buf, 'B', 1, e))}, vcc")

    def rescale(self, clear=True):
        """O *= alpha (per row; 1 exactly where the row max stayed inside its headroom)."""
        self.i("s_nop 15")
        self.i("s_nop 15")                    # last PV MFMA -> v_accvgpr_read
        for X in "AB":
            al = vr(STV(X, 'al'))
            for b0 in range(0, 16 * self.DB, 4):
                regs = [self.OA(X, 0) + b0 + k for k in range(4)]
                for k, a in enumerate(regs):
                    self.i(f"v_accvgpr_read_b32 {vr(V_E[k])}, a{a}")
                for k in range(4):
                    self.i(f"v_mul_f32 {vr(V_E[k])}, {vr(V_E[k])}, {al}")
                for k, a in enumerate(regs):
                    self.i(f"v_accvgpr_write_b32 a{a}, {vr(V_E[k])}")
        self.i("s_nop 7")
        if clear:
            self.i(f"s_mov_b64 {S('grow')}, 0")

    # ---- item decode: unit counter / sub item -> bases, descriptors, tile count (SALU only) ----------------------------------------
    def decode(self):
        """In: n_u = unit counter i of the item to decode, n_sub its sub item (causal: 0 heavy block, 1 light block).
        Out: ksrd_n / vsrd_n (bases), qsrd_n, n_nt, n_qblk, n_b, n_hh, n_valid.  A counter past the workgroup's list leaves the
        descriptors as they are (the stream then re-fetches tiles of a valid item: harmless) and gives n_valid = 0, n_nt = maxint."""
        u, t0, t1, t3 = S('t3'), S('t0'), S('t1'), S('t2', 0)
        th = S('t2', 1)
        lv, ln = self.ul("valid"), self.ul("decoded")
        # u = slot + SL * i  (the units of this XCD's heads, head-major, dealt round-robin over its CUs: a head's units run on
        # consecutive CUs at the same time)
        # (slot / xcd of this workgroup: xcd_mode ? (xcd = wg & 7, slot = wg >> 3) : (xcd = 0, slot = wg); the workgroup id stays in s2)
        self.i(f"s_lshr_b32 {t0}, s2, 3")
        self.i(f"s_cmp_eq_u32 {ka('xcd_mode')}, 0")
        self.i(f"s_cselect_b32 {t0}, s2, {t0}")                           # slot
        self.i(f"s_mul_i32 {u}, {S('n_u')}, {ka('SL')}")
        self.i(f"s_add_u32 {u}, {u}, {t0}")
        self.i(f"s_mul_i32 {t0}, {ka('hx')}, {ka('NU')}")                 # the units of this XCD's heads
        self.i(f"s_cmp_lt_u32 {u}, {t0}")
        self.i(f"s_cbranch_scc1 {lv}")
        self.i(f"s_mov_b32 {S('n_nt')}, 0x7fffffff")
        self.i(f"s_mov_b32 {S('n_valid')}, 0")
        self.i(f"s_branch {ln}")
        self.lab(lv)
        self.i(f"s_mov_b32 {S('n_valid')}, 1")
        # lh = u / NU, p = u % NU
        self.i(f"s_mul_hi_u32 {t0}, {u}, {ka('magic_NU')}")
        self.i(f"s_cmp_eq_u32 {ka('NU')}, 1")
        self.i(f"s_cselect_b32 {t0}, {u}, {t0}")                       # lh
        self.i(f"s_mul_i32 {t1}, {t0}, {ka('NU')}")
        self.i(f"s_sub_u32 {t1}, {u}, {t1}")                           # p
        if self.causal:
            self.i(f"s_sub_u32 {t3}, {ka('NB')}, 1")
            self.i(f"s_sub_u32 {t3}, {t3}, {t1}")                      # heavy block NB-1-p
            self.i(f"s_cmp_eq_u32 {S('n_sub')}, {1 if LIGHT1 else 0}")
            self.i(f"s_cselect_b32 {t1}, {t3}, {t1}")                  # qblk (LIGHT1: a unit's light block first, then its heavy one)
            self.i(f"s_add_u32 {t3}, {t1}, 1")
            self.i(f"s_lshl_b32 {S('n_nt')}, {t3}, 2")                 # nt = 4 (qblk + 1)
        else:
            self.i(f"s_mov_b32 {S('n_nt')}, {ka('nt_full')}")
        self.i(f"s_mov_b32 {S('n_qblk')}, {t1}")
        # bh = xcd_mode ? xcd + 8 lh : lh
        self.i(f"s_lshl_b32 {t3}, {t0}, 3")
        self.i(f"s_and_b32 {th}, s2, 7")                               # xcd
        self.i(f"s_add_u32 {t3}, {t3}, {th}")
        self.i(f"s_cmp_eq_u32 {ka('xcd_mode')}, 0")
        self.i(f"s_cselect_b32 {t0}, {t0}, {t3}")                      # bh
        # b = bh / H, hh = bh % H, hkv = hh / kv_group
        self.i(f"s_mul_hi_u32 {t1}, {t0}, {ka('magic_H')}")
        self.i(f"s_cmp_eq_u32 {ka('H')}, 1")
        self.i(f"s_cselect_b32 {t1}, {t0}, {t1}")                      # b
        self.i(f"s_mul_i32 {t3}, {t1}, {ka('H')}")
        self.i(f"s_sub_u32 {t0}, {t0}, {t3}")                          # hh
        self.i(f"s_mov_b32 {S('n_b')}, {t1}")
        self.i(f"s_mov_b32 {S('n_hh')}, {t0}")
        self.i(f"s_mul_hi_u32 {u}, {t0}, {ka('magic_G')}")
        self.i(f"s_cmp_eq_u32 {ka('kv_group')}, 1")
        self.i(f"s_cselect_b32 {u}, {t0}, {u}")                        # hkv
        # K / V bases: ptr + b * sb + hkv * sh (byte strides, 32-bit)
        for nm, sr in (('k', 'ksrd_n'), ('v', 'vsrd_n')):
            lo, hi = S(sr, 0), S(sr, 1)
            self.i(f"s_mul_i32 {lo}, {t1}, {ka(nm + '_sb')}")
            self.i(f"s_mul_hi_u32 {hi}, {t1}, {ka(nm + '_sb')}")
            self.i(f"s_mul_i32 {t3}, {u}, {ka(nm + '_sh')}")
            self.i(f"s_mul_hi_u32 {th}, {u}, {ka(nm + '_sh')}")
            self.i(f"s_add_u32 {lo}, {lo}, {t3}")
            self.i(f"s_addc_u32 {hi}, {hi}, {th}")
            self.i(f"s_add_u32 {lo}, {lo}, {ka(nm)}")
            self.i(f"s_addc_u32 {hi}, {hi}, {ka(nm, hi=True)}")
            self.i(f"s_and_b32 {hi}, {hi}, 0xffff")
        # Q descriptor: base = ptr + b * sb + hh * sh + qblk * 256 * ss  (records / flags are set once in the kernel prologue)
        lo, hi = S('qsrd_n', 0), S('qsrd_n', 1)
        self.i(f"s_mul_i32 {lo}, {t1}, {ka('q_sb')}")
        self.i(f"s_mul_hi_u32 {hi}, {t1}, {ka('q_sb')}")
        self.i(f"s_mul_i32 {t3}, {t0}, {ka('q_sh')}")
        self.i(f"s_mul_hi_u32 {th}, {t0}, {ka('q_sh')}")
        self.i(f"s_add_u32 {lo}, {lo}, {t3}")
        self.i(f"s_addc_u32 {hi}, {hi}, {th}")
        self.i(f"s_lshl_b32 {u}, {S('n_qblk')}, 8")
        self.i(f"s_mul_i32 {t3}, {u}, {ka('q_ss')}")
        self.i(f"s_mul_hi_u32 {th}, {u}, {ka('q_ss')}")
        self.i(f"s_add_u32 {lo}, {lo}, {t3}")
        self.i(f"s_addc_u32 {hi}, {hi}, {th}")
        self.i(f"s_add_u32 {lo}, {lo}, {ka('q')}")
        self.i(f"s_addc_u32 {hi}, {hi}, {ka('q', hi=True)}")
        self.i(f"s_and_b32 {hi}, {hi}, 0xffff")
        if self.klen or self.kmask:            # records = the block's rows that exist
            self.block_records(S('qsrd_n', 2), S('n_qblk'), 'q_ss', self.RB, t3)
        if (self.klen or self.kmask) and not self.causal:
            # seqlens_k (kernarg dwords 40 / 41, null = none): the item's keys L = clamp(seqlens_k[b], 0, Sk) and its tile count cut to them
            # (an even number of tiles, at least 4): a padded batch does not compute its padding.  klen: L also feeds the length words.
            lno = self.ul("noseqlens")
            self.i(f"s_mov_b32 {S('L_n')}, {ka('Sk')}")
            self.i(f"s_load_dwordx2 {S('t2')}, s[0:1], {4 * KA_SEQLENS}")
            self.i("s_waitcnt lgkmcnt(0)")
            self.i(f"s_cmp_eq_u64 {S('t2')}, 0")
            self.i(f"s_cbranch_scc1 {lno}")
            self.i(f"s_lshl_b32 {t0}, {S('n_b')}, 2")
            self.i(f"s_load_dword {t0}, {S('t2')}, {t0}")
            self.i("s_waitcnt lgkmcnt(0)")
            self.i(f"s_max_i32 {t0}, {t0}, 0")
            self.i(f"s_min_i32 {S('L_n')}, {t0}, {ka('Sk')}")
            self.i(f"s_add_u32 {t0}, {S('L_n')}, 127")
            self.i(f"s_lshr_b32 {t0}, {t0}, 7")
            self.i(f"s_lshl_b32 {t0}, {t0}, 1")
            self.i(f"s_max_u32 {t0}, {t0}, 4")
            self.i(f"s_min_u32 {S('n_nt')}, {t0}, {ka('nt_full')}")
            self.lab(lno)
        if self.cl:
            # seqlens_k under the causal mask (the padded decoder batch): with L = clamp(seqlens_k[b], 0, Sk) keys and G = max(1, ceil(L / 256))
            # 256-key groups holding any of them, block qblk either keeps its diagonal (qblk + 1 <= G: the causal item as ever, the
            # length words mask what lies past L inside its last group) or lies wholly behind the cut (qblk + 1 > G): then it runs the 4 G
            # tiles of those groups like a non-causal item -- every wave the same tiles, no diagonal (nd_n = NODIAG).  Nothing past the
            # group of the last visible key is fetched or computed.
            lno = self.ul("noseqlens")
            self.i(f"s_mov_b32 {self.Lcut}, {ka('Sk')}")
            self.i(f"s_mov_b32 {self.nd_n}, 0")
            self.i(f"s_load_dwordx2 {S('t2')}, s[0:1], {4 * KA_SEQLENS}")
            self.i("s_waitcnt lgkmcnt(0)")
            self.i(f"s_cmp_eq_u64 {S('t2')}, 0")
            self.i(f"s_cbranch_scc1 {lno}")
            self.i(f"s_lshl_b32 {t0}, {S('n_b')}, 2")
            self.i(f"s_load_dword {t0}, {S('t2')}, {t0}")
            self.i("s_waitcnt lgkmcnt(0)")
            self.i(f"s_max_i32 {t0}, {t0}, 0")
            self.i(f"s_min_i32 {self.Lcut}, {t0}, {ka('Sk')}")
            self.i(f"s_add_u32 {t0}, {self.Lcut}, 255")
            self.i(f"s_lshr_b32 {t0}, {t0}, 8")
            self.i(f"s_max_u32 {t0}, {t0}, 1")                          # G
            self.i(f"s_add_u32 {t1}, {S('n_qblk')}, 1")
            self.i(f"s_cmp_gt_u32 {t1}, {t0}")
            self.i(f"s_cselect_b32 {self.nd_n}, {NODIAG}, 0")
            self.i(f"s_min_u32 {t0}, {t0}, {t1}")
            self.i(f"s_lshl_b32 {S('n_nt')}, {t0}, 2")
            self.lab(lno)
        self.lab(ln)

    def block_records(self, dst, qblk, ss, row_bytes, tmp):
        """dst = (min(256, Sq - 256 qblk) - 1) * stride + row bytes   (klen: descriptors that end with the block's last existing row)"""
        self.i(f"s_lshl_b32 {tmp}, {qblk}, 8")
        self.i(f"s_sub_u32 {tmp}, {ka('Sq')}, {tmp}")
        self.i(f"s_min_u32 {tmp}, {tmp}, 256")
        self.i(f"s_sub_u32 {tmp}, {tmp}, 1")
        if isinstance(ss, str):
            self.i(f"s_mul_i32 {tmp}, {tmp}, {ka(ss)}")
        else:
            self.i(f"s_mul_i32 {tmp}, {tmp}, {ss}")
        self.i(f"s_add_u32 {dst}, {tmp}, {row_bytes}")

    def make_out_srds(self):
        """osrd / lsrd of the item that becomes current, from n_b / n_hh / n_qblk (before the next decode overwrites them)."""
        t0, t1, t3, th = S('t0'), S('t1'), S('t2', 0), S('t2', 1)
        lo, hi = S('osrd', 0), S('osrd', 1)
        self.i(f"s_mul_i32 {lo}, {S('n_b')}, {ka('o_sb')}")
        self.i(f"s_mul_hi_u32 {hi}, {S('n_b')}, {ka('o_sb')}")
        self.i(f"s_mul_i32 {t3}, {S('n_hh')}, {ka('o_sh')}")
        self.i(f"s_mul_hi_u32 {th}, {S('n_hh')}, {ka('o_sh')}")
        self.i(f"s_add_u32 {lo}, {lo}, {t3}")
        self.i(f"s_addc_u32 {hi}, {hi}, {th}")
        self.i(f"s_lshl_b32 {t0}, {S('n_qblk')}, 8")
        self.i(f"s_mul_i32 {t3}, {t0}, {ka('o_ss')}")
        self.i(f"s_mul_hi_u32 {th}, {t0}, {ka('o_ss')}")
        self.i(f"s_add_u32 {lo}, {lo}, {t3}")
        self.i(f"s_addc_u32 {hi}, {hi}, {th}")
        self.i(f"s_add_u32 {lo}, {lo}, {ka('o')}")
        self.i(f"s_addc_u32 {hi}, {hi}, {ka('o', hi=True)}")
        self.i(f"s_and_b32 {hi}, {hi}, 0xffff")
        # lse: base = lse + ((b * H + hh) * Sq + qblk * 256) * 4
        self.i(f"s_mul_i32 {t1}, {S('n_b')}, {ka('H')}")
        self.i(f"s_add_u32 {t1}, {t1}, {S('n_hh')}")
        self.i(f"s_mul_i32 {t3}, {t1}, {ka('Sq')}")
        self.i(f"s_mul_hi_u32 {th}, {t1}, {ka('Sq')}")
        self.i(f"s_add_u32 {t3}, {t3}, {t0}")
        self.i(f"s_addc_u32 {th}, {th}, 0")
        self.i(f"s_lshl_b64 {S('t2')}, {S('t2')}, 2")
        self.i(f"s_add_u32 {S('lsrd', 0)}, {ka('lse')}, {t3}")
        self.i(f"s_addc_u32 {S('lsrd', 1)}, {ka('lse', hi=True)}, {th}")
        self.i(f"s_and_b32 {S('lsrd', 1)}, {S('lsrd', 1)}, 0xffff")
        if self.klen or self.kmask:            # both end with the block's last existing row
            self.block_records(S('osrd', 2), S('n_qblk'), 'o_ss', self.RB * (2 if self.out32 else 1), t3)
            self.block_records(S('lsrd', 2), S('n_qblk'), 4, 4, t3)

    def zero_o(self):
        """O = 0 on the matrix pipe: one MFMA of zero operands per 16 accumulator registers (8 instructions instead of 128 writes)."""
        z = V_T[0]
        for k in range(4):
            self.i(f"v_mov_b32 {vr(z + k)}, 0")
        self.i("s_nop 3")                     # VALU write -> MFMA operand read
        for b0 in range(0, 2 * 16 * self.DB, 16):
            self.i(f"{self.mf} {ar(b0, 16)}, {vr(z, 4)}, {vr(z, 4)}, 0")

    # ---- per-item prologue and epilogue ------------------------------------------------------------------------------------------
    def q_fragments(self, lg=None):
        """This wave's 64 rows out of its landing zone into the accumulator file (per-lane addresses recomputed per item: nothing
        lane-constant is kept live for it); lg: the LDS-read tracker of the phase that follows."""
        qb = V_E[0:8]
        T0, T1, T2 = (vr(x) for x in V_T[0:3])
        # address = qland + r * 256 + (((2 ks + h) ^ (r & 15)) << 4)  =  rowbase ^ (32 ks),  qland = Q_BASE + wave * TILE
        self.i(f"s_lshl_b32 {S('t0')}, {S('w4k')}, 2")
        self.i(f"s_add_u32 {S('t0')}, {S('t0')}, {self.Q_BASE}")
        if self.D == 128:
            self.i(f"v_and_b32 {T0}, 31, {vr(V_LANE)}")                               # r
            self.i(f"v_lshrrev_b32 {T1}, 5, {vr(V_LANE)}")                            # h
            self.i(f"v_and_b32 {T2}, 15, {T0}")
            self.i(f"v_xor_b32 {T2}, {T2}, {T1}")
            self.i(f"v_lshlrev_b32 {T2}, 4, {T2}")
            self.i(f"v_lshl_add_u32 {T2}, {T0}, 8, {T2}")
            self.i(f"v_add_u32 {T2}, {S('t0')}, {T2}")
            for ks in range(8):
                self.i(f"v_xor_b32 {vr(qb[ks])}, {32 * ks}, {T2}")
        else:                                  # D = 64: a strip's 32 rows lie in the landing zone exactly as 32 keys lie in a K tile image
            for ks in range(self.KS):
                self.i(f"v_add_u32 {vr(qb[ks])}, {S('t0')}, {vr(KOFF(ks))}")
        for X, off in (('A', 0), ('B', self.HALF)):
            for ks in range(self.KS):
                self.i(f"ds_read_b128 {ar(self.QA(X, ks), 4)}, {vr(qb[ks])} offset:{off}")
                if lg is not None:
                    lg.issue(('q', X, ks))

    def state_reset(self, emit=True):
        o = []
        if self.fast:                          # (m c comes from the item's tile 0: start_fast with_max; STV thr holds the constant 2^14)
            for X in "AB":
                o += [f"v_mov_b32 {vr(STV(X, 'l'))}, 0", f"v_mov_b32 {vr(STV(X, 'ps0'))}, 0", f"v_mov_b32 {vr(PS1(X))}, 0",
                      f"v_mov_b32 {vr(STV(X, 'al'))}, 1.0"]
            if emit:
                self.emit(o)
            return o
        for X in "AB":
            o += [f"v_mov_b32 {vr(STV(X, 'm'))}, {NEG_BIG}",
                  f"v_mov_b32 {vr(STV(X, 'thr'))}, {NEG_BIG}",
                  f"v_mul_f32 {vr(STV(X, 'mc'))}, {ka('scale_log2')}, {vr(STV(X, 'm'))}",
                  f"v_mov_b32 {vr(STV(X, 'l'))}, 0",
                  f"v_mov_b32 {vr(STV(X, 'ps0'))}, 0",
                  f"v_mov_b32 {vr(STV(X, 'al'))}, 1.0"]
        o.append(f"v_mov_b32 {vr(V_PS1)}, 0")
        if PKADD and not self.split:
            for X in "AB":
                o += [f"v_mov_b32 {vr(PSP(X))}, 0", f"v_mov_b32 {vr(PSP(X) + 1)}, 0"]
        if emit:
            self.emit(o)
        return o

    def tile0_diag(self, cond):
        """Causal: tile 0 is this wave's diagonal tile when its tile count is 1; `cond` = instructions that leave SCC = 1 in that case."""
        if not self.causal:
            return
        lm, lr = self.ul("mask0"), self.ul("masked0")
        self.emit(cond)
        self.i(f"s_cbranch_scc1 {lm}")
        self.lab(lr)
        self.out_of_line(True)
        self.lab(lm)
        self.mask_diag(0)
        self.i(f"s_branch {lr}")
        self.out_of_line(False)

    def body_seam(self, full):
        """The last iteration of an item that has a successor, parity 1: the K/V stream already delivers the successor's tiles 0 / 1
        and its Q rows landed iterations ago, so this IS a FULL iteration once the Q fragments are swapped -- QK^T_next(0) beside
        finish(j), PV(j) beside start_next(0) -- instead of a LAST body, an epilogue and a prologue with nothing to overlap.  The
        ending item's row sums / maxima move to SV() for its epilogue, which follows; O is zeroed after that (zero_o).
        full: this wave's last tile is j (else it is past its last tile: only the successor's half of the work)."""
        p = 1
        self.cm(f"SEAM body ({'last tile' if full else 'past the last tile'}): QK^T_next(0) || finish(j);  PV(j) || start_next(0)")
        park = []
        for X in "AB":
            park += [f"v_mov_b32 {vr(SV(X, 'l'))}, {vr(STV(X, 'l'))}", f"v_mov_b32 {vr(SV(X, 'mc'))}, {vr(STV(X, 'mc'))}"]
        park += self.state_reset(emit=False)
        if self.fast:
            sta = self.start_fast('A', 0, with_max=True) + self.start_fast('B', 0, with_max=True)
        else:
            sta = self.start_stream('A', 0) + self.start_stream('B', 0)
        unit = [f"v_mov_b32 {vr(STV(X, 'al'))}, 1.0" for X in "AB"]
        # tile 0 of the successor is this wave's diagonal tile iff its tile count is 1: nt_n - 3 + wave == 1
        cond = [f"s_add_u32 {S('t0')}, {S('nt_n')}, {S('wave')}"] + ([f"s_add_u32 {S('t0')}, {S('t0')}, {self.nd_n}"] if self.cl else []) + \
               [f"s_cmp_eq_u32 {S('t0')}, 4"]
        lg = self.kprefetch_lg() if self.mb else Lgkm()      # (mid-barrier loop: K_next(0)'s first fragments were requested an iteration ago)
        if full:
            self.abl_on = True
            self.q_fragments(lg)
            if self.fast:
                fin = self.interleave(self.finish_fast('A', p), self.finish_fast('B', p, alt=True)) + self.tile_check()
            else:
                fin = self.finish_stream('A', p) + self.finish_stream('B', p)
            n = 2 * self.NKF
            gv = [g * n // 32 for g in DMA_GAPS_V][:self.PPW] if self.PPW == 4 else [1 * n // 16, 5 * n // 16]
            gk = [g * n // 32 for g in DMA_GAPS_K][:self.PPW] if self.PPW == 4 else [9 * n // 16, 13 * n // 16]
            dma = self.dma_plan(p, gv, gk)
            lg = self.run_phase(self.phase_qk, fin, lg, p=p, dma_at=dma, tail_vreads=p)
            if self.fast:                      # the ending item's last tile: its check (VCC is tile_check's), then its sums into l before l is parked
                self.fix_check(p)
                park = self.lupd('A') + self.lupd('B') + park
            self.mask_keys(0, 0)
            self.tile0_diag(cond)
            self.run_phase(self.phase_pv, park + sta + unit, lg, p=p, dma_at={}, preissued=True)
            self.abl_on = False
        else:
            self.i(self.setm0('v', 1 - p))
            self.i("s_nop 0")
            for t in range(self.PPW):
                self.i(self.dma('V', t))
            self.i(self.setm0('k', p))
            self.i("s_nop 0")
            for t in range(self.PPW):
                self.i(self.dma('K', t))
            self.q_fragments(lg)
            self.emit(park)
            self.phase_qk(p, [[] for _ in range(2 * self.NKF)], {}, lg=lg)
            self.mask_keys(0, 0)
            self.tile0_diag(cond)
            self.i("s_nop 7")
            self.i("s_nop 7")
            self.emit(sta + unit)
        self.i(f"s_mov_b64 {S('grow')}, 0")

    def item_prologue(self):
        """Q fragments out of the landing zone into the accumulator file, state, QK^T(0) (O zeroed in its shadow), softmax start(0)."""
        self.cm("item prologue")
        # this item's Q pieces: everything but the previous item's output stores (issued after them: 16, or 32 of the fp32 epilogue)
        self.i(f"s_waitcnt vmcnt({(8 if self.out32 else 4) * self.DB})")
        self.q_fragments()
        self.state_reset()
        self.i("s_waitcnt lgkmcnt(0)")
        # QK^T(0) from K slot 0 into buffer 0; four O zeros per gap
        self.zero_o()
        self.phase_qk(1, [[] for _ in range(2 * self.NKF)], {})       # parity argument 1: target buffer 0, K slot 0
        self.mask_keys(0, 0)
        self.tile0_diag([f"s_cmp_eq_u32 {S('wrem')}, 0"])          # wnt == 1
        self.i("s_nop 7")
        self.i("s_nop 7")
        for X in "AB":
            self.emit(self.start_fast(X, 0, with_max=True) if self.fast else self.start_stream(X, 0))
        for X in "AB":
            self.i(f"v_mov_b32 {vr(STV(X, 'al'))}, 1.0")
        self.i(f"s_mov_b64 {S('grow')}, 0")

    def item_epilogue(self, saved=False):
        """Normalise, convert, stage the strip as a [32 rows][256 B] image in the wave's own LDS, store whole rows; LSE.
        saved: behind a pipelined seam (body_seam) -- the item's row sums / maxima are in SV(), S buffer 0 and the softmax state already
        belong to the next item."""
        self.cm("item epilogue" + (" (behind a pipelined seam)" if saved else ""))
        self.i("s_nop 15")
        self.i("s_nop 15")                    # last MFMA -> v_accvgpr_read
        ST_ = SV if saved else STV
        if self.out32:
            return self.item_epilogue_f32(ST_)
        vb, rb = V_E[8], V_E[9:13]            # staging write base, read-back bases
        inv, lt, t = V_E[13], V_E[14], V_E[15]
        L, T0, T1, T2, T3 = vr(V_LANE), *(vr(x) for x in V_T)
        gofs = V_PS1
        W = S('wave')
        self.i(f"s_lshl_b32 {S('t0')}, {S('w4k')}, 1")
        self.i(f"s_add_u32 {S('t0')}, {S('t0')}, {self.O_BASE}")                  # ostage = O_BASE + wave * TILE / 2
        self.i(f"v_and_b32 {T0}, 31, {L}")
        self.i(f"v_lshrrev_b32 {T1}, 5, {L}")
        if self.D == 128:
            # write base: ostage + r * 256 + ((h ^ (r & 15)) << 4)   (chunk 4 db + g + h lands at ((4 db + g) << 4) ^ that)
            self.i(f"v_and_b32 {T2}, 15, {T0}")
            self.i(f"v_xor_b32 {T2}, {T2}, {T1}")
            self.i(f"v_lshlrev_b32 {T2}, 4, {T2}")
            self.i(f"v_lshl_add_u32 {T2}, {T0}, 8, {T2}")
            self.i(f"v_add_u32 {vr(vb)}, {S('t0')}, {T2}")
        else:
            # D = 64: the strip is staged as 32 keys lie in a K tile image (two rows per 256 B, same swizzle): chunk 4 db + g + h of row r
            # sits where K fragment k-step 2 db + g / 2 is read from
            self.i(f"v_add_u32 {vr(vb)}, {S('t0')}, {vr(KOFF(0))}")
        # LSE offset of the lane's row inside the item: (64 wave + r) * 4
        self.i(f"s_lshl_b32 {S('t1')}, {W}, 8")
        self.i(f"v_lshlrev_b32 {T3}, 2, {T0}")
        self.i(f"v_add_u32 {T3}, {S('t1')}, {T3}")
        if self.D == 128:
            # read-back bases: ostage + (4 k + q4) * 256 + ((c16 ^ (4 k + q4)) << 4), k = 0..3; rows 16 further: + 4096
            self.i(f"v_lshrrev_b32 {T0}, 4, {L}")                                   # q4
            self.i(f"v_and_b32 {T1}, 15, {L}")                                      # c16
            for k in range(4):
                self.i(f"v_add_u32 {T2}, {4 * k}, {T0}")                              # row
                self.i(f"v_xor_b32 {vr(rb[k])}, {T1}, {T2}")
                self.i(f"v_lshlrev_b32 {vr(rb[k])}, 4, {vr(rb[k])}")
                self.i(f"v_lshl_add_u32 {vr(rb[k])}, {T2}, 8, {vr(rb[k])}")
                self.i(f"v_add_u32 {vr(rb[k])}, {S('t0')}, {vr(rb[k])}")
        else:
            # read-back: lane (q8 = lane >> 3, c8 = lane & 7) takes chunk c8 of row 8 k + q8, k = 0..3: LDS row R = 4 k + (q8 >> 1),
            # position (((q8 & 1) << 3) | c8) ^ sw(R), sw(R) = ((q8 >> 1) << 2) | k
            self.i(f"v_lshrrev_b32 {T0}, 3, {L}")                                   # q8
            self.i(f"v_and_b32 {T1}, 15, {L}")                                      # ((q8 & 1) << 3) | c8
            self.i(f"v_lshrrev_b32 {T2}, 4, {L}")                                   # q8 >> 1
            for k in range(4):
                self.i(f"v_lshl_or_b32 {vr(rb[k])}, {T2}, 2, {k}")                    # sw
                self.i(f"v_xor_b32 {vr(rb[k])}, {vr(rb[k])}, {T1}")
                self.i(f"v_lshlrev_b32 {vr(rb[k])}, 4, {vr(rb[k])}")
                self.i(f"v_add_u32 {vr(t)}, {4 * k}, {T2}")                           # R
                self.i(f"v_lshl_add_u32 {vr(rb[k])}, {vr(t)}, 8, {vr(rb[k])}")
                self.i(f"v_add_u32 {vr(rb[k])}, {S('t0')}, {vr(rb[k])}")
            self.i(f"v_and_b32 {T1}, 7, {L}")                                       # c8
        # store offset inside the item's output rows: (64 wave + q) * o_ss + 16 c
        self.i(f"s_lshl_b32 {S('t1')}, {W}, 6")                                    # 64 wave
        self.i(f"v_add_u32 {T0}, {S('t1')}, {T0}")
        self.i(f"v_mul_lo_u32 {vr(gofs)}, {T0}, {ka('o_ss')}")
        self.i(f"v_lshl_add_u32 {vr(gofs)}, {T1}, 4, {vr(gofs)}")
        RPI = 1024 // self.RB                                                     # rows one store instruction covers
        self.i(f"s_lshl_b32 {S('t1')}, {ka('o_ss')}, {2 if RPI == 4 else 3}")
        for X in "AB":
            l, mc = vr(ST_(X, 'l')), vr(ST_(X, 'mc'))
            self.i(f"v_mov_b32 {vr(t)}, {l}")
            self.i("s_nop 1")
            self.i(f"v_permlane32_swap_b32 {l}, {vr(t)}")
            self.i(f"v_add_f32 {vr(lt)}, {l}, {vr(t)}")
            self.i(f"v_rcp_f32 {vr(inv)}, {vr(lt)}")
            self.i(f"v_cmp_lt_f32 vcc, 0, {vr(lt)}")
            self.i("s_nop 1")
            self.i(f"v_cndmask_b32 {vr(inv)}, 0, {vr(inv)}, vcc")
            w = V_E[0:8]
            for db in range(self.DB):
                for g in (0, 2):
                    for k in range(8):
                        self.i(f"v_accvgpr_read_b32 {vr(w[k])}, a{self.OA(X, db, 4 * g + k)}")
                    for k in range(8):
                        self.i(f"v_mul_f32 {vr(w[k])}, {vr(w[k])}, {vr(inv)}")
                    # ua = (w0 w1 | w2 w3), ub = (w4 w5 | w6 w7) -> registers w0 w1 (ua) and w2 w3 (ub) after conversion
                    self.i(f"{self.cvt} {vr(w[0])}, {vr(w[0])}, {vr(w[1])}")
                    self.i(f"{self.cvt} {vr(w[1])}, {vr(w[2])}, {vr(w[3])}")
                    self.i(f"{self.cvt} {vr(w[2])}, {vr(w[4])}, {vr(w[5])}")
                    self.i(f"{self.cvt} {vr(w[3])}, {vr(w[6])}, {vr(w[7])}")
                    self.i(f"v_xor_b32 {vr(w[4])}, {(4 * db + g) << 4}, {vr(vb)}")
                    self.i("s_nop 0")
                    self.i(f"v_permlane32_swap_b32 {vr(w[0])}, {vr(w[2])}")
                    self.i(f"v_permlane32_swap_b32 {vr(w[1])}, {vr(w[3])}")
                    self.i(f"ds_write_b128 {vr(w[4])}, {vr(w[0], 4)}")
            # LSE = (m c + log2(l)) ln 2, stored by the lower lane half (one lane per row)
            self.i(f"v_log_f32 {vr(t)}, {vr(lt)}")
            self.i(f"v_mov_b32 {vr(inv)}, {NEG_INF}")
            self.i(f"v_add_f32 {vr(t)}, {vr(t)}, {mc}")
            self.i(f"v_mul_f32 {vr(t)}, 0x3f317218, {vr(t)}")
            self.i(f"v_cndmask_b32 {vr(t)}, {vr(inv)}, {vr(t)}, vcc")
            ll = self.ul("nolse")
            self.i(f"s_cmp_eq_u64 {ka('lse', 2)}, 0")
            self.i(f"s_cbranch_scc1 {ll}")
            self.i("s_mov_b32 exec_hi, 0")
            self.i(f"buffer_store_dword {vr(t)}, {T3}, {S('lsrd')}, 0 offen offset:{0 if X == 'A' else 128}")
            self.i("s_mov_b32 exec_hi, -1")
            self.lab(ll)
            self.i("s_waitcnt lgkmcnt(0)")
            NI = 32 // RPI
            x = [84 + 4 * k for k in range(NI)]                 # S buffer 1 is dead here (an item's last tile is odd): the 16-byte read-back registers
            for k in range(NI):
                self.i(f"ds_read_b128 {vr(x[k], 4)}, {vr(rb[k & 3])} offset:{(k >> 2) * 4096}")
            if X == 'A':
                self.i(f"s_mov_b32 {S('t0')}, 0")
            else:
                self.i(f"s_lshl_b32 {S('t0')}, {ka('o_ss')}, 5")                   # strip B: 32 rows further
            for k in range(NI):
                self.i(f"s_waitcnt lgkmcnt({NI - 1 - k})")
                if not (saved and "nostore" in ABL):      # (TIMING-ONLY ablation: the seam path's epilogue without its output stores)
                    self.i(f"buffer_store_dwordx4 {vr(x[k], 4)}, {vr(gofs)}, {S('osrd')}, {S('t0')} offen")
                if k < NI - 1:
                    self.i(f"s_add_u32 {S('t0')}, {S('t0')}, {S('t1')}")
        self.i(f"v_mov_b32 {vr(V_PS1)}, 0")

    def item_epilogue_f32(self, ST_=STV):
        """Parity variant: normalise and store fp32 straight from the accumulators (a lane holds 4 consecutive columns of its row per
        register quad: one 16-byte store each); LSE as in the 16-bit epilogue."""
        inv, lt, t = V_E[13], V_E[14], V_E[15]
        L, T0, T1, T2, T3 = vr(V_LANE), *(vr(x) for x in V_T)
        gofs = V_PS1
        self.i(f"v_and_b32 {T0}, 31, {L}")                                          # r
        self.i(f"v_lshrrev_b32 {T1}, 5, {L}")                                       # h
        self.i(f"v_mul_lo_u32 {vr(gofs)}, {T0}, {ka('o_ss')}")
        self.i(f"v_lshl_add_u32 {vr(gofs)}, {T1}, 4, {vr(gofs)}")                   # r * o_ss + 16 h
        self.i(f"s_lshl_b32 {S('t1')}, {S('wave')}, 8")
        self.i(f"v_lshlrev_b32 {T3}, 2, {T0}")
        self.i(f"v_add_u32 {T3}, {S('t1')}, {T3}")                                  # LSE offset (64 wave + r) * 4
        self.i(f"s_lshl_b32 {S('t1')}, {S('wave')}, 6")                             # 64 wave
        self.i(f"s_mul_i32 {S('t1')}, {S('t1')}, {ka('o_ss')}")                     # the wave's first output row
        for X in "AB":
            l, mc = vr(ST_(X, 'l')), vr(ST_(X, 'mc'))
            self.i(f"v_mov_b32 {vr(t)}, {l}")
            self.i("s_nop 1")
            self.i(f"v_permlane32_swap_b32 {l}, {vr(t)}")
            self.i(f"v_add_f32 {vr(lt)}, {l}, {vr(t)}")
            self.i(f"v_rcp_f32 {vr(inv)}, {vr(lt)}")
            self.i(f"v_cmp_lt_f32 vcc, 0, {vr(lt)}")
            self.i("s_nop 1")
            self.i(f"v_cndmask_b32 {vr(inv)}, 0, {vr(inv)}, vcc")
            if X == 'B':
                self.i(f"s_lshl_b32 {S('t0')}, {ka('o_ss')}, 5")
                self.i(f"s_add_u32 {S('t1')}, {S('t1')}, {S('t0')}")                # strip B: 32 rows further
            n = 0
            for db in range(self.DB):
                for g in range(4):
                    w = V_E[4 * (n & 1):4 * (n & 1) + 4]     # two register quads in turn: a store's data is not overwritten right behind it
                    n += 1
                    for k in range(4):
                        self.i(f"v_accvgpr_read_b32 {vr(w[k])}, a{self.OA(X, db, 4 * g + k)}")
                    for k in range(4):
                        self.i(f"v_mul_f32 {vr(w[k])}, {vr(w[k])}, {vr(inv)}")
                    self.i(f"buffer_store_dwordx4 {vr(w[0], 4)}, {vr(gofs)}, {S('osrd')}, {S('t1')} offen offset:{128 * db + 32 * g}")
            self.i(f"v_log_f32 {vr(t)}, {vr(lt)}")
            self.i(f"v_mov_b32 {vr(inv)}, {NEG_INF}")
            self.i(f"v_add_f32 {vr(t)}, {vr(t)}, {mc}")
            self.i(f"v_mul_f32 {vr(t)}, 0x3f317218, {vr(t)}")
            self.i(f"v_cndmask_b32 {vr(t)}, {vr(inv)}, {vr(t)}, vcc")
            ll = self.ul("nolse")
            self.i(f"s_cmp_eq_u64 {ka('lse', 2)}, 0")
            self.i(f"s_cbranch_scc1 {ll}")
            self.i("s_mov_b32 exec_hi, 0")
            self.i(f"buffer_store_dword {vr(t)}, {T3}, {S('lsrd')}, 0 offen offset:{0 if X == 'A' else 128}")
            self.i("s_mov_b32 exec_hi, -1")
            self.lab(ll)
        self.i(f"v_mov_b32 {vr(V_PS1)}, 0")

    def lane_constants_d128(self, L, T0, T1, T2, T3, W):
        """LDS read addresses and DMA source offsets, D = 128 (one key per 256-B LDS row).  T0 = r, T1 = h on entry."""
        # koff[ks] = r*256 + (((2ks+h) ^ sw(r)) << 4), sw(r) = ((r&3)<<2) | ((r>>2)&3)  ==  base ^ (32 ks)
        self.i(f"v_and_b32 {T2}, 3, {T0}")
        self.i(f"v_lshlrev_b32 {T2}, 2, {T2}")
        self.i(f"v_bfe_u32 {T3}, {T0}, 2, 2")
        self.i(f"v_or_b32 {T2}, {T2}, {T3}")
        self.i(f"v_xor_b32 {T2}, {T2}, {T1}")
        self.i(f"v_lshlrev_b32 {T2}, 4, {T2}")
        self.i(f"v_lshl_add_u32 {T2}, {T0}, 8, {T2}")
        for ks in range(8):
            self.i(f"v_xor_b32 {vr(KOFF(ks))}, {32 * ks}, {T2}")
        # voff[db][hi] = V_BASE + R*256 + ((ch ^ sw(R)) << 4) + 8 (tp&1);  R = 4h + tq + 8hi, ch = 4db + 2 g1 + (tp>>1),
        # sw(R) = (tq << 2) | (h + 2 hi);  g1 = (lane>>4)&1, tq = (lane&15)>>2, tp = lane&3
        e0, e1, e2, e3, e4 = (vr(x) for x in V_E[1:6])
        self.i(f"v_bfe_u32 {e0}, {L}, 2, 2")                                      # tq
        self.i(f"v_and_b32 {e1}, 3, {L}")                                         # tp
        self.i(f"v_bfe_u32 {e2}, {L}, 4, 1")                                      # g1
        self.i(f"v_lshrrev_b32 {e3}, 1, {e1}")
        self.i(f"v_lshl_add_u32 {e3}, {e2}, 1, {e3}")                             # c2 = 2 g1 + (tp >> 1)
        self.i(f"v_and_b32 {e1}, 1, {e1}")
        self.i(f"v_lshlrev_b32 {e1}, 3, {e1}")                                    # 8 (tp & 1)
        self.i(f"v_lshl_add_u32 {e4}, {T1}, 2, {e0}")                             # 4h + tq
        for hi in range(2):
            for db in range(4):
                d = vr(VOFF(db, hi))
                # ch ^ sw = ((db ^ tq) << 2) | (c2 ^ (h + 2 hi))
                self.i(f"v_xor_b32 {T2}, {db}, {e0}")
                self.i(f"v_add_u32 {T3}, {2 * hi}, {T1}")
                self.i(f"v_xor_b32 {T3}, {T3}, {e3}")
                self.i(f"v_lshl_add_u32 {T2}, {T2}, 2, {T3}")
                self.i(f"v_lshlrev_b32 {T2}, 4, {T2}")
                self.i(f"v_add_u32 {T3}, {8 * hi}, {e4}")                         # R
                self.i(f"v_lshl_add_u32 {T2}, {T3}, 8, {T2}")
                self.i(f"v_add_u32 {T2}, {T2}, {e1}")
                self.i(f"v_add_u32 {d}, {self.V_BASE}, {T2}")
        # DMA source offsets: row R = 16 wave + 4 t + q4 -> R * ss + ((c16 ^ ((q4 << 2) | t)) << 4) - 1024 t
        self.i(f"v_lshrrev_b32 {e0}, 4, {L}")                                     # q4
        self.i(f"v_and_b32 {e1}, 15, {L}")                                        # c16
        self.i(f"s_lshl_b32 {S('t0')}, {W}, 4")
        self.i(f"s_lshl_b32 {S('t1')}, {W}, 6")
        for t in range(4):
            self.i(f"v_lshl_or_b32 {e2}, {e0}, 2, {t}")
            self.i(f"v_xor_b32 {e2}, {e2}, {e1}")
            self.i(f"v_lshlrev_b32 {e2}, 4, {e2}")
            self.i(f"v_add_u32 {e3}, {4 * t}, {e0}")
            self.i(f"v_add_u32 {e3}, {S('t0')}, {e3}")                            # R
            for d, ss in ((KDOFF(t), 'k_ss'), (VDOFF(t), 'v_ss')):
                self.i(f"v_mul_lo_u32 {e4}, {e3}, {ka(ss)}")
                self.i(f"v_add_u32 {e4}, {e4}, {e2}")
                self.i(f"v_subrev_u32 {vr(d)}, {1024 * t}, {e4}")
            # Q: row-in-16 = 4 t + q4 -> (64 wave + 4t + q4) * q_ss + ((c16 ^ (4t + q4)) << 4) - 1024 t
            self.i(f"v_add_u32 {e3}, {4 * t}, {e0}")
            self.i(f"v_xor_b32 {e2}, {e3}, {e1}")
            self.i(f"v_lshlrev_b32 {e2}, 4, {e2}")
            self.i(f"v_add_u32 {e3}, {S('t1')}, {e3}")                            # + the wave's first row of the block (64 wave)
            self.i(f"v_mul_lo_u32 {e4}, {e3}, {ka('q_ss')}")
            self.i(f"v_add_u32 {e4}, {e4}, {e2}")
            self.i(f"v_subrev_u32 {vr(QDOFF(t))}, {1024 * t}, {e4}")

    def tile_off64(self, dst, key, ch, t0, t1):
        """dst = byte offset of 16-byte chunk ch of key `key` in a D = 64 tile image: two keys per 256-B LDS row R = key >> 1 (the odd
        key in the upper half), position (((key & 1) << 3) | ch) ^ sw(R), sw(R) = ((R & 3) << 2) | ((R >> 2) & 3).  key / ch / dst / t0 /
        t1: VGPR names; dst may be key or ch."""
        self.i(f"v_lshrrev_b32 {t0}, 1, {key}")                                   # R
        self.i(f"v_and_b32 {t1}, 1, {key}")
        self.i(f"v_lshl_or_b32 {t1}, {t1}, 3, {ch}")                              # c
        self.i(f"v_and_b32 {dst}, 3, {t0}")
        self.i(f"v_lshlrev_b32 {dst}, 2, {dst}")
        self.i(f"v_xor_b32 {t1}, {t1}, {dst}")
        self.i(f"v_bfe_u32 {dst}, {t0}, 2, 2")
        self.i(f"v_xor_b32 {t1}, {t1}, {dst}")                                    # c ^ sw
        self.i(f"v_lshlrev_b32 {t1}, 4, {t1}")
        self.i(f"v_lshl_add_u32 {dst}, {t0}, 8, {t1}")

    def lane_constants_d64(self, L, T0, T1, T2, T3, W):
        """LDS read addresses and DMA source offsets, D = 64 (two keys per 256-B LDS row, see tile_off64).  T0 = r, T1 = h on entry."""
        e0, e1, e2, e3, e4, e5, e6 = (vr(x) for x in V_E[1:8])
        # koff[ks] = off(r, 2 ks + h) = off(r, h) ^ (32 ks)
        self.tile_off64(T2, T0, T1, T3, e0)
        for ks in range(self.KS):
            self.i(f"v_xor_b32 {vr(KOFF(ks))}, {32 * ks}, {T2}")
        # voff[2 s2 + db][hi] = V_BASE + off(16 s2 + 8 hi + 4 h + tq, 4 db + 2 g1 + (tp >> 1)) + 8 (tp & 1)
        self.i(f"v_bfe_u32 {e0}, {L}, 2, 2")                                      # tq
        self.i(f"v_and_b32 {e1}, 3, {L}")                                         # tp
        self.i(f"v_bfe_u32 {e2}, {L}, 4, 1")                                      # g1
        self.i(f"v_lshrrev_b32 {e3}, 1, {e1}")
        self.i(f"v_lshl_add_u32 {e3}, {e2}, 1, {e3}")                             # c2 = 2 g1 + (tp >> 1)
        self.i(f"v_and_b32 {e1}, 1, {e1}")
        self.i(f"v_lshlrev_b32 {e1}, 3, {e1}")                                    # 8 (tp & 1)
        self.i(f"v_lshl_add_u32 {e4}, {T1}, 2, {e0}")                             # 4h + tq
        for s2 in range(2):
            for db in range(2):
                for hi in range(2):
                    d = vr(VOFF(2 * s2 + db, hi))
                    self.i(f"v_add_u32 {e5}, {16 * s2 + 8 * hi}, {e4}")           # key
                    self.i(f"v_add_u32 {e6}, {4 * db}, {e3}")                     # chunk
                    self.tile_off64(e5, e5, e6, T2, T3)
                    self.i(f"v_add_u32 {e5}, {e5}, {e1}")
                    self.i(f"v_add_u32 {d}, {self.V_BASE}, {e5}")
        # DMA source offsets.  A piece is 1 KiB = 4 LDS rows; lane (q4 = lane >> 4, c16 = lane & 15) fills position c16 of LDS row
        # Rl = 8 wave + 4 t + q4 of the image, which holds c = c16 ^ sw(Rl), sw = (q4 << 2) | ((2 wave + t) & 3): chunk c & 7 of key
        # 2 Rl + (c >> 3)  ->  key * ss + ((c & 7) << 4) - 1024 t  (the instruction's immediate adds 1024 t to both addresses)
        self.i(f"v_lshrrev_b32 {e0}, 4, {L}")                                     # q4
        self.i(f"v_and_b32 {e1}, 15, {L}")                                        # c16
        for t in range(self.PPW):
            self.i(f"s_lshl_b32 {S('t0')}, {W}, 1")
            self.i(f"s_add_u32 {S('t0')}, {S('t0')}, {t}")
            self.i(f"s_and_b32 {S('t0')}, {S('t0')}, 3")
            self.i(f"v_lshl_or_b32 {e2}, {e0}, 2, {S('t0')}")                     # sw
            self.i(f"v_xor_b32 {e2}, {e2}, {e1}")                                 # c
            self.i(f"s_lshl_b32 {S('t0')}, {W}, 3")
            self.i(f"s_add_u32 {S('t0')}, {S('t0')}, {4 * t}")
            self.i(f"v_add_u32 {e3}, {S('t0')}, {e0}")                            # Rl
            self.i(f"v_lshrrev_b32 {e4}, 3, {e2}")
            self.i(f"v_lshl_add_u32 {e3}, {e3}, 1, {e4}")                         # key
            self.i(f"v_and_b32 {e2}, 7, {e2}")
            self.i(f"v_lshlrev_b32 {e2}, 4, {e2}")                                # 16 (c & 7)
            for d, ss in ((KDOFF(t), 'k_ss'), (VDOFF(t), 'v_ss')):
                self.i(f"v_mul_lo_u32 {e4}, {e3}, {ka(ss)}")
                self.i(f"v_add_u32 {e4}, {e4}, {e2}")
                self.i(f"v_subrev_u32 {vr(d)}, {1024 * t}, {e4}")
        # Q: a group of four pieces is one strip (16 LDS rows = 32 query rows): piece t, lane (q4, c16) -> LDS row Rl = 4 t + q4,
        # c = c16 ^ ((q4 << 2) | t): chunk c & 7 of row 64 wave + 2 Rl + (c >> 3) of the block (the second group: qoff + 32 rows)
        self.i(f"s_lshl_b32 {S('t1')}, {W}, 6")
        for t in range(4):
            self.i(f"v_lshl_or_b32 {e2}, {e0}, 2, {t}")
            self.i(f"v_xor_b32 {e2}, {e2}, {e1}")                                 # c
            self.i(f"v_add_u32 {e3}, {4 * t}, {e0}")                              # Rl
            self.i(f"v_lshrrev_b32 {e4}, 3, {e2}")
            self.i(f"v_lshl_add_u32 {e3}, {e3}, 1, {e4}")                         # row in the strip
            self.i(f"v_add_u32 {e3}, {S('t1')}, {e3}")
            self.i(f"v_and_b32 {e2}, 7, {e2}")
            self.i(f"v_lshlrev_b32 {e2}, 4, {e2}")
            self.i(f"v_mul_lo_u32 {e4}, {e3}, {ka('q_ss')}")
            self.i(f"v_add_u32 {e4}, {e4}, {e2}")
            self.i(f"v_subrev_u32 {vr(QDOFF(t))}, {1024 * t}, {e4}")

    def item_switch(self):
        # ---- item switch: the decoded item becomes current; decode the one after it ----------------------------------------------------
        self.i(f"s_mov_b32 {S('irem')}, {S('n_nt')}")
        self.i(f"s_sub_u32 {S('krem')}, {S('n_nt')}, 2")
        self.i(f"s_sub_u32 {S('vrem')}, {S('n_nt')}, 1")
        if self.cl:                            # behind the cut every wave runs every tile (wnt = nt); `nd` = what wrem equals one iteration before the diagonal tile
            self.i(f"s_cmp_eq_u32 {self.nd_n}, 0")
            self.i(f"s_cselect_b32 {S('wrem')}, {S('wave')}, 3")
            self.i(f"s_cselect_b32 {self.nd}, 1, 0x80000000")
            self.i(f"s_add_u32 {S('wrem')}, {S('wrem')}, {S('n_nt')}")
            self.i(f"s_sub_u32 {S('wrem')}, {S('wrem')}, 4")
        elif self.causal:
            self.i(f"s_add_u32 {S('wrem')}, {S('n_nt')}, {S('wave')}")
            self.i(f"s_sub_u32 {S('wrem')}, {S('wrem')}, 4")                      # wnt - 1 = nt - 4 + wave
        else:
            self.i(f"s_sub_u32 {S('wrem')}, {S('n_nt')}, 1")
        self.make_out_srds()
        if self.causal:
            # the next item: the unit's light block after its heavy one, else the next unit's heavy block.  With an odd number of
            # blocks the last unit is the middle block alone (heavy == light: 2 qblk + 1 == NB): no second item.
            lone, ldone = self.ul("loneblock"), self.ul("advanced")
            self.i(f"s_lshl_b32 {S('t0')}, {S('n_qblk')}, 1")
            self.i(f"s_add_u32 {S('t0')}, {S('t0')}, 1")
            self.i(f"s_sub_u32 {S('t0')}, {S('t0')}, {ka('NB')}")
            self.i(f"s_or_b32 {S('t0')}, {S('t0')}, {S('n_sub')}")               # 0 <=> the middle block, as a heavy (first) item
            self.i(f"s_cmp_eq_u32 {S('t0')}, 0")
            self.i(f"s_cbranch_scc1 {lone}")
            self.i(f"s_xor_b32 {S('n_sub')}, {S('n_sub')}, 1")
            self.i(f"s_cmp_eq_u32 {S('n_sub')}, 0")
            self.i(f"s_cselect_b32 {S('t0')}, 1, 0")
            self.i(f"s_add_u32 {S('n_u')}, {S('n_u')}, {S('t0')}")
            self.i(f"s_branch {ldone}")
            self.lab(lone)
            self.i(f"s_add_u32 {S('n_u')}, {S('n_u')}, 1")
            self.lab(ldone)
        else:
            self.i(f"s_add_u32 {S('n_u')}, {S('n_u')}, 1")
        self.decode()
        self.i(f"s_mov_b32 {S('nt_n')}, {S('n_nt')}")
        self.i(f"s_lshl_b32 {S('qrem')}, {S('n_valid')}, {2 if self.PPW == 4 else 1}")   # PPW groups of Q pieces if there is a next item
        self.i(f"s_lshl_b32 {S('qdst')}, {S('w4k')}, 2")
        self.i(f"s_add_u32 {S('qdst')}, {S('qdst')}, {self.Q_BASE}")
        self.i(f"s_mov_b32 {S('qoff')}, 0")

    # ---- the kernel --------------------------------------------------------------------------------------------------------------
    def kernel(self):
        n = self.name
        self.L += [f"\t.globl\t{n}", "\t.p2align\t8", f"\t.type\t{n},@function", f"{n}:"]
        self.i(f"s_load_dwordx16 s[{KBASE_SGPR}:{KBASE_SGPR + 15}], s[0:1], 0")
        self.i(f"s_load_dwordx16 s[{KBASE_SGPR + 16}:{KBASE_SGPR + 31}], s[0:1], 64")
        self.i(f"s_load_dwordx8 s[{KBASE_SGPR + 32}:{KBASE_SGPR + 39}], s[0:1], 128")
        L, T0, T1, T2, T3 = vr(V_LANE), *(vr(x) for x in V_T)
        W = S('wave')
        self.i(f"v_and_b32 {L}, 63, v0")
        self.i(f"v_lshrrev_b32 {T0}, 6, v0")
        self.i("s_nop 0")
        self.i(f"v_readfirstlane_b32 {W}, {T0}")
        self.i("s_waitcnt lgkmcnt(0)")
        # slot / xcd of this workgroup: xcd_mode ? (xcd = wg & 7, slot = wg >> 3) : (xcd = 0, slot = wg)
        self.stamp_init()
        self.i(f"s_lshl_b32 {S('ktile')}, {ka('k_ss')}, 6")
        self.i(f"s_lshl_b32 {S('vtile')}, {ka('v_ss')}, 6")
        self.i(f"s_lshl_b32 {S('w4k')}, {W}, {12 if self.D == 128 else 11}")         # the wave's quarter of a tile image
        # constant descriptor words: records = bytes of one (batch, head) slab / of one item's rows, flags = raw buffer
        for sr, ss, rows in (('ksrd', 'k_ss', 'Sk'), ('vsrd', 'v_ss', 'Sk'), ('qsrd_n', 'q_ss', None), ('osrd', 'o_ss', None)):
            if rows:
                self.i(f"s_sub_u32 {S('t0')}, {ka(rows)}, 1")
                self.i(f"s_mul_i32 {S('t0')}, {S('t0')}, {ka(ss)}")
            else:
                self.i(f"s_mul_i32 {S('t0')}, {ka(ss)}, 255")
            self.i(f"s_add_u32 {S(sr, 2)}, {S('t0')}, {self.RB * (2 if (sr == 'osrd' and self.out32) else 1)}")
            self.i(f"s_mov_b32 {S(sr, 3)}, 0x00020000")
        self.i(f"s_mov_b32 {S('lsrd', 2)}, 1024")
        self.i(f"s_mov_b32 {S('lsrd', 3)}, 0x00020000")
        # -- lane constants
        self.i(f"v_and_b32 {T0}, 31, {L}")                                        # r
        self.i(f"v_lshrrev_b32 {T1}, 5, {L}")                                     # h
        # causal threshold tA = r + 1 - 4h
        self.i(f"v_lshlrev_b32 {T2}, 2, {T1}")
        self.i(f"v_sub_u32 {vr(V_TA)}, {T0}, {T2}")
        self.i(f"v_add_u32 {vr(V_TA)}, 1, {vr(V_TA)}")
        if self.D == 128:
            self.lane_constants_d128(L, T0, T1, T2, T3, W)
        else:
            self.lane_constants_d64(L, T0, T1, T2, T3, W)
        if self.fast and not self.kmask:       # (key-mask kernels: start_fast sets a row's limit with its item's tile 0)
            for X in "AB":
                self.i(f"v_mov_b32 {vr(STV(X, 'thr'))}, 0x46800000")              # 2^14: the tile-sum limit of the fast loop (finish_fast)
        # -- first item: decode, fetch its Q block, K0, then V0 and K1
        self.i(f"s_mov_b32 {S('n_u')}, 0")
        self.i(f"s_mov_b32 {S('n_sub')}, 0")
        self.decode()
        lend = f".L{n}_end"
        self.i(f"s_cmp_eq_u32 {S('n_valid')}, 0")
        self.i(f"s_cbranch_scc1 {lend}")
        self.i(f"s_lshl_b32 {S('qdst')}, {S('w4k')}, 2")
        self.i(f"s_add_u32 {S('qdst')}, {S('qdst')}, {self.Q_BASE}")
        self.i(f"s_mov_b32 {S('qoff')}, 0")
        if self.kmask:                         # mask bytes of tiles 0 and 1, ahead of every DMA piece (the counted wait below covers them)
            self.i(f"s_mul_i32 {ka('pad')}, {S('n_b')}, {ka('Sk')}")
            self.emit(self.mask_load(0))
            self.i(f"s_add_u32 {ka('pad')}, {ka('pad')}, 64")
            self.emit(self.mask_load(1))
            self.i(f"s_add_u32 {ka('pad')}, {ka('pad')}, 64")                     # next: tile 2
        if self.klen:      # the words of tiles 0 and 1; keys left from tile 2 on
            self.i(f"s_mov_b32 {ka('pad')}, {S('L_n')}")
            self.len_word(0)
            self.i(f"s_sub_u32 {ka('pad')}, {ka('pad')}, 64")
            self.len_word(1)
            self.i(f"s_sub_u32 {ka('pad')}, {ka('pad')}, 64")
        self.i(f"s_mov_b32 {S('qrem')}, {self.PPW}")
        for g in range(self.PPW):                # the wave's 64 rows, 4 KiB a group
            self.q_group()
        self.i(f"s_mov_b64 {S('ksrd', 0, 2)}, {S('ksrd_n')}")
        self.i(f"s_mov_b64 {S('vsrd', 0, 2)}, {S('vsrd_n')}")
        self.i(f"s_mov_b32 {S('koff')}, 0")
        self.i(f"s_mov_b32 {S('voff')}, 0")
        self.i(self.setm0('k', 0))
        self.i("s_nop 0")
        for t in range(self.PPW):
            self.i(self.dma('K', t))
        self.i(self.setm0('v', 0))
        self.i("s_nop 0")
        for t in range(self.PPW):
            self.i(self.dma('V', t))
        self.i(f"s_add_u32 {S('koff')}, {S('koff')}, {S('ktile')}")
        self.i(self.setm0('k', 1))
        self.i("s_nop 0")
        for t in range(self.PPW):
            self.i(self.dma('K', t))
        self.i(f"s_add_u32 {S('koff')}, {S('koff')}, {S('ktile')}")               # next K piece: tile 2
        self.i(f"s_add_u32 {S('voff')}, {S('voff')}, {S('vtile')}")               # next V piece: tile 1
        self.i(f"s_waitcnt vmcnt({2 * self.PPW})")                                # Q and K0 (the pieces of V0, K1 are the youngest)
        if self.kmask:
            self.i(self.mask_word(0, 0))
            self.i(self.mask_word(1, 1))
            self.i(f"s_mov_b32 {self.kleft}, {self.kleft0}")
            self.km_len_and(0)
            self.km_len_and(1)
        self.i("s_barrier")
        litem = f".L{n}_item"
        self.lab(litem)
        self.item_switch()
        self.item_prologue()
        self.i("s_waitcnt vmcnt(0)")
        self.i("s_barrier")                    # V0 / K1 published; every wave is done with K slot 0
        if self.mb:
            self.emit(self.kprefetch(1))       # the loop's first iteration (parity 0) reads K(1) from slot 1
        self.stamp(3, count=8)
        # ---- tile loop, unrolled by the two S buffers ----------------------------------------------------------------------------------
        lloop, lgen = f".L{n}_loop", f".L{n}_generic"
        self.lab(lloop)
        if LEAN and STAMP in (0, 3):
            # Two tiles at a time WITHOUT the per-iteration bookkeeping, while this wave is far from both ends of the item: no Q
            # pieces left to request (qrem == 0), no stream switch in either iteration (krem >= 2) and, under the causal mask, FULL
            # bodies without a diagonal tile (wrem = krem - 2 + wave >= 3).  A lean iteration issues the same DMA pieces and meets the
            # same barriers as a generic one, so every wave decides for itself.  (17 scalar instructions a tile were ~9 % of it.)
            # (lengths: tiles j+2, j+3 whole too; a causal item's last block may run three tiles past the keys: its mask bytes)
            self.i(f"s_cmp_lt_u32 {S('krem')}, {(7 if (self.kmask or self.klen) else 5) if self.causal else (4 if (self.klen or self.kmask) else 2)}")
            self.i(f"s_cbranch_scc1 {lgen}")
            self.i(f"s_cmp_lg_u32 {S('qrem')}, 0")
            self.i(f"s_cbranch_scc1 {lgen}")
            for p in (0, 1):
                if self.kmask:
                    self.emit(self.mask_load())
                self.body_full(p, lean=True)
                if self.mb:                         # (offsets, counted wait and barrier sit inside the body's PV phase)
                    continue
                self.i(f"s_add_u32 {S('koff')}, {S('koff')}, {S('ktile')}")
                self.i(f"s_add_u32 {S('voff')}, {S('voff')}, {S('vtile')}")
                if self.kmask:
                    self.i(f"s_add_u32 {ka('pad')}, {ka('pad')}, 64")
                self.i("s_waitcnt vmcnt(0)")
                if self.kmask:
                    self.i(self.mask_word(p))
                self.i("s_barrier")
            for c in ("krem", "vrem", "wrem", "irem"):
                self.i(f"s_sub_u32 {S(c)}, {S(c)}, 2")
            if self.kmask:                          # (their words need no length: all keys exist)
                self.i(f"s_sub_u32 {self.kleft}, {self.kleft}, 128")
            if self.klen:       # the words of tiles j+2, j+3 (all keys exist), the keys left behind them
                self.i(f"s_mov_b64 {self.MK(0)}, -1")
                self.i(f"s_mov_b64 {self.MK(1)}, -1")
                self.i(f"s_sub_u32 {ka('pad')}, {ka('pad')}, 128")
            self.i(f"s_branch {lloop}")
            self.lab(lgen)
        use_seam = SEAM and STAMP in (0, 3)
        lseam = f".L{n}_seam"
        for p in (0, 1):
            lnf, ll, ld = (self.ul(x) for x in ("notfull", "last", "done"))
            if p == 1 and use_seam:           # the item's last iteration, and another item follows: the pipelined seam (below)
                lnorm = self.ul("noseam")
                self.i(f"s_cmp_eq_u32 {S('irem')}, 2")
                self.i(f"s_cbranch_scc0 {lnorm}")
                self.i(f"s_cmp_lg_u32 {S('n_valid')}, 0")
                self.i(f"s_cbranch_scc1 {lseam}")
                self.lab(lnorm)
            self.stream_top(p)
            self.i(f"s_cmp_gt_i32 {S('wrem')}, 0")
            self.i(f"s_cbranch_scc0 {lnf}")
            self.body_full(p)
            self.lab(ld)
            if not self.mb:
                self.stream_bottom(p)
            self.out_of_line(True)
            self.lab(lnf)
            self.i(f"s_cmp_eq_u32 {S('wrem')}, 0")
            self.i(f"s_cbranch_scc1 {ll}")
            self.body_skip(p)
            if self.mb:                             # (no PV phase to carry the barrier: at the end, the K fragments of the next iteration behind it)
                self.stream_bottom(p)
                self.emit(self.kprefetch(p))
            self.i(f"s_branch {ld}")
            self.lab(ll)
            self.body_last(p)
            if self.mb:
                self.stream_bottom(p)
                self.emit(self.kprefetch(p))
            self.i(f"s_branch {ld}")
            self.out_of_line(False)
        self.i(f"s_sub_u32 {S('irem')}, {S('irem')}, 2")
        self.i(f"s_cmp_gt_i32 {S('irem')}, 0")
        self.i(f"s_cbranch_scc1 {lloop}")
        self.stamp(2)
        self.item_epilogue()
        self.stamp(4)
        self.i(f"s_cmp_lg_u32 {S('n_valid')}, 0")
        self.i(f"s_cbranch_scc1 {litem}")
        self.stamp_dump()
        self.lab(lend)
        self.i("s_endpgm")
        if use_seam:
            # ---- the pipelined seam: last iteration of an item with a successor, the ending item's epilogue, O = 0, item switch ----------
            lsk, lsd = self.ul("seamskip"), self.ul("seamdone")
            self.lab(lseam)
            self.stream_top(1)
            self.i(f"s_cmp_eq_u32 {S('wrem')}, 0")
            self.i(f"s_cbranch_scc0 {lsk}")
            self.body_seam(True)
            self.i(f"s_branch {lsd}")
            self.lab(lsk)
            self.body_seam(False)
            self.lab(lsd)
            self.stream_bottom(1)
            if self.mb:
                self.emit(self.kprefetch(1))   # the next item's iteration 0 (parity 0) reads K_next(1) from slot 1
            if "noepi" not in ABL:             # (TIMING-ONLY ablations of the seam path, wrong results: what an item costs outside its tile loop)
                self.item_epilogue(saved=True)
            if "nozero" not in ABL:
                self.zero_o()
            self.item_switch()
            self.i(f"s_branch {lloop}")
        assert not self.lstack and self.L is self.main
        self.main += self.ool
        self.main += self.ool2
        if self.fast:                          # the fix-up subroutines, behind every call site (fix_check adds a positive offset to its pc)
            self.L = self.main
            for b in (0, 1):
                self.lab(f".L{n}_fixup{b}")
                self.fixup(b)
        self.main += [f".L{n}_fend:", f"\t.size\t{n}, .L{n}_fend-{n}"]

    def descriptor(self):
        n = self.name
        return f"""
	.section	.rodata,"a",@progbits
	.p2align	6, 0x0
	.amdhsa_kernel {n}
		.amdhsa_group_segment_fixed_size {self.LDS_BYTES}
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_kernarg_size {4 * KARG_MEM_DWORDS}
		.amdhsa_user_sgpr_count 2
		.amdhsa_user_sgpr_kernarg_segment_ptr 1
		.amdhsa_system_sgpr_workgroup_id_x 1
		.amdhsa_system_sgpr_workgroup_id_y 0
		.amdhsa_system_sgpr_workgroup_id_z 0
		.amdhsa_system_vgpr_workitem_id 0
		.amdhsa_next_free_vgpr 512
		.amdhsa_next_free_sgpr {KBASE_SGPR + KARG_DWORDS + 2}
		.amdhsa_accum_offset 256
		.amdhsa_reserve_vcc 1
		.amdhsa_float_round_mode_32 0
		.amdhsa_float_round_mode_16_64 0
		.amdhsa_float_denorm_mode_32 3
		.amdhsa_float_denorm_mode_16_64 3
		.amdhsa_dx10_clamp 1
		.amdhsa_ieee_mode 1
		.amdhsa_fp16_overflow 0
		.amdhsa_tg_split 0
	.end_amdhsa_kernel
	.text
"""

    def metadata(self):
        n = self.name
        return f"""  - .agpr_count:     256
    .args:
      - .offset:         0
        .size:           {4 * KARG_MEM_DWORDS}
        .value_kind:     by_value
    .group_segment_fixed_size: {self.LDS_BYTES}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {4 * KARG_MEM_DWORDS}
    .max_flat_workgroup_size: 256
    .name:           {n}
    .private_segment_fixed_size: 0
    .sgpr_count:     {KBASE_SGPR + KARG_DWORDS + 2 + 6}
    .sgpr_spill_count: 0
    .symbol:         {n}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     512
    .vgpr_spill_count: 0
    .wavefront_size: 64
"""


def kernels():
    """(dtype, causal, parity): parity = fp32 store + split P (the <= 1e-3 variant on the same schedule)"""
    return [(dt, D, causal, km, par) for dt in ("bf16", "fp16") for D in (128, 64) for causal in (True, False)
            for km in (("",) if STAMP else ("", "km", "kl")) for par in (False, True)]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--offsets":       # the kernarg layout as a C header fragment (pfa_p4.hip static_asserts it)
        for k, v in KA.items():
            print(f"#define P4_KA_{k.upper()} {4 * v}")
        print(f"#define P4_KARG_BYTES {4 * KARG_MEM_DWORDS}")
        print(f"#define P4_KA_SEQLENS {4 * KA_SEQLENS}")
        print(f"#define P4_LDS_BYTES_D128 {10 * 128 * 128}")
        print(f"#define P4_LDS_BYTES_D64 {10 * 128 * 64}")
        return
    out = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6", "\t.text"]
    meta = []
    for dt, D, causal, km, par in kernels():
        g = Gen(dt, causal, out32=par, split=par, D=D, kmask=km == "km", klen=km == "kl")
        g.kernel()
        out += g.main
        out.append(g.descriptor())
        meta.append(g.metadata())
    out.append("\t.amdgpu_metadata\n---\namdhsa.kernels:\n" + "".join(meta) +
               "amdhsa.target:   amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\t.end_amdgpu_metadata")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
