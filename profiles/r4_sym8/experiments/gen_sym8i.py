#!/usr/bin/env python3
"""Generate quantized-cnn_amd/csrc/qcnn_sym8i_pos.h: one asm statement per output position of k_conv_sym8i that does BOTH
jobs of a wave in a stage period — its share of the next stage's table build (matrix instructions + add-TID stores) and the
look-ups of the position in the current stage — interleaved instruction by instruction.

Why.  In k_conv_sym8 a wave multiplies, then gathers: measured (scripts/trace_sym8.py) 38 % of a period in the build phase,
8 % issuing operand loads, 49 % in the look-ups, and a stage costs 2540 + 1.97 x look-ups cycles — the two phases add up.
A gather wave spends most of its look-up phase WAITING for LDS round trips (two waves per SIMD cannot cover them for each
other); the matrix pipe is idle meanwhile.  Here every position's look-up stream carries a slice of the build in those gaps:
after the reads of a block are issued, a matrix instruction (or a tile's four stores) goes out before the wave waits for the
block issued before.  The operands of the build come from LDS (the kernel DMAs each stage's code-book block and activation
rows there a period ahead), so no operand lives in a register across the period and no vector load is issued by the wave.

Text per (channels per wave CPW, positions NP, position index P, k-steps KS): which of the wave's eight result tiles the
position builds is static, so are the registers: look-up temporaries v[240:255] (two sets of two reads), offset slots
v[236:239], below them code-book operands (4 per k-step), activation operands and two result tiles in flight (RegMap).
Every s_waitcnt is a COUNT of the LDS operations issued after the one waited for (LDS returns in order; stores count too),
computed by simulating the issue order.  A position whose taps do not cover the stage's pixel runs the build slice alone.
usage: python scripts/gen/gen_sym8i.py > quantized-cnn_amd/csrc/qcnn_sym8i_pos.h
"""
import os
RB, NSETS = 2, 2
VAR = int(os.environ.get("S8I_VAR", "0"))     # timing experiments only: 1 no stores, 2 no matrix instructions, 4 no padding nops, 8 stores at the end
T_BASE = [240, 248]                 # look-up temporaries of the two sets
SLOT = [(236, 237), (238, 239)]     # offset slots
TILES = [(t, i) for t in range(2) for i in range(4)]


class RegMap:
    """Fixed registers of one text, packed downwards from v239: code-book operands (4 per k-step), activation operands (one per
    k-step and image tile the position builds for), two result tiles."""

    def __init__(self, KS, ts):
        self.A = [236 - 4 * KS + 4 * ks for ks in range(KS)]
        nb = KS * len(ts)
        b0 = self.A[0] - nb
        self.B = {(t, ks): b0 + n * KS + ks for n, t in enumerate(ts) for ks in range(KS)}
        r1 = (b0 - 4) & ~1                      # register tuples start at even registers
        self.R = [r1 - 4, r1]
        self.first = r1 - 4


def tiles_of(p, NP):
    return TILES[(8 * p) // NP:(8 * (p + 1)) // NP]


class Stream:
    def __init__(self):
        self.lines, self.issued, self.last_mfma = [], [], {}

    def a(self, s):
        self.lines.append(s)

    def mark_mfma(self, d):
        self.last_mfma[d] = len(self.lines)

    def lines_since_mfma(self, d):
        return len(self.lines) - self.last_mfma[d]

    def lds(self, name, text):
        self.lines.append(text)
        self.issued.append(name)

    def wait_for(self, name):
        n = len(self.issued) - 1 - self.issued.index(name)
        assert 0 <= n <= 15, (name, n)
        self.lines.append('s_waitcnt lgkmcnt(%d)' % n)

    def wait_all(self):
        self.lines.append('s_waitcnt lgkmcnt(0)')


def filler_units(st, tiles, KS, tag, rm):
    """[callable] units of the build slice: operand reads, then per tile its matrix instructions, its stores two units later."""
    ts = sorted({t for t, _ in tiles})
    A_BASE, R_BASE = rm.A, rm.R
    B_REG = lambda t, ks: rm.B[(t, ks)]

    def load():
        for ks in range(KS):
            st.lds('%sA%d' % (tag, ks), 'ds_read_b128 v[%d:%d], %%[opA] offset:%%[oa%d]' % (A_BASE[ks], A_BASE[ks] + 3, ks))
        for t in ts:
            for ks in range(KS):
                st.lds('%sB%d_%d' % (tag, t, ks), 'ds_read_b32 v%d, %%[opB] offset:%%[ob%d_%d]' % (B_REG(t, ks), t, ks))

    def mfma(j):
        t, i = tiles[j]
        d = R_BASE[j & 1]

        def run():
            if j == 0:       # operands: the last one issued is the last one to land
                st.wait_for('%sB%d_%d' % (tag, ts[-1], KS - 1))
            if not (VAR & 2):
                st.a('v_mfma_f32_16x16x4_f32 v[%d:%d], v%d, v%d, 0' % (d, d + 3, A_BASE[0] + i, B_REG(t, 0)))
                if KS > 1:
                    st.a('v_mfma_f32_16x16x4_f32 v[%d:%d], v%d, v%d, v[%d:%d]' % (d, d + 3, A_BASE[1] + i, B_REG(t, 1), d, d + 3))
            st.mark_mfma(d)
        return run

    def store(j, last):
        t, i = tiles[j]
        d = R_BASE[j & 1]

        def run():
            # A store reads the tile's result registers: the matrix instructions (8 passes = 32 cycles each, the second k-step
            # behind the first) must have retired.  Behind a LATER tile's matrix instructions they have (those could not
            # issue before); a single k-step leaves a few cycles of write-back; the last tile needs real padding unless
            # enough look-up instructions (>= 4 cycles each) were issued since.
            if VAR & 1:
                return
            if last and st.lines_since_mfma(d) < 20 and not (VAR & 4):
                for _ in range(4):
                    st.a('s_nop 15')
            elif KS == 1:
                st.a('s_nop 7')
            st.a('s_mov_b32 m0, %%[m%d]' % t)
            st.a('s_nop 0')
            for e in range(4):
                st.lds('%sS%d_%d' % (tag, j, e), 'ds_write_addtid_b32 v%d offset:%d' % (d + e, i * 1024 + e * 256))
        return run

    n = len(tiles)
    units = [load]
    for j in range(n):
        units.append(mfma(j))
        if j >= 1:
            units.append(store(j - 1, False))
    units.append(store(n - 1, True))
    return units


def emit(CPW, NP, P, KS):
    reads = CPW // 2
    NB, NG = reads // RB, reads // 4
    tiles = tiles_of(P, NP)
    rm = RegMap(KS, sorted({t for t, _ in tiles}))
    # ---------------- valid path: look-ups with the build slice in the gaps
    st = Stream()
    units = filler_units(st, tiles, KS, 'v', rm)
    units[0]()                                          # operand reads first: they land under the first blocks
    rest = units[1:]
    slot_of = [min(NB - 1, (u * NB) // len(rest)) for u in range(len(rest))]
    if VAR & 8:      # experiment: the whole build slice behind the look-ups
        slot_of = [NB for _ in rest]

    def fetch(g):
        lo, hi = SLOT[g & 1]
        st.lds('F%d' % g, 'ds_read_b64 v[%d:%d], %%[blk] offset:%%[i%d]' % (lo, hi, g))

    def accumulate(k):
        base = T_BASE[k % NSETS]
        for r in range(RB):
            st.wait_for('R%d_%d' % (k, r))
            c = 2 * (k * RB + r)
            st.a('v_pk_add_f32 %%[c%d], v[%d:%d], %%[c%d]' % (c, base + 4 * r, base + 4 * r + 1, c))
            st.a('v_pk_add_f32 %%[c%d], v[%d:%d], %%[c%d]' % (c + 1, base + 4 * r + 2, base + 4 * r + 3, c + 1))

    fetch(0)
    if NG > 1:
        fetch(1)
    for k in range(NB):
        base = T_BASE[k % NSETS]
        first_read = k * RB
        g = first_read // 4
        if first_read % 4 == 0:
            st.wait_for('F%d' % g)
        for r in range(RB):
            q = (first_read + r) % 4
            src = SLOT[g & 1][q >> 1]
            st.a('v_xor_b32_sdwa v%d, v%d, %%[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
                 % (base + 4 * r, src, q & 1))
        for r in range(RB):
            st.lds('R%d_%d' % (k, r), 'ds_read_b128 v[%d:%d], v%d' % (base + 4 * r, base + 4 * r + 3, base + 4 * r))
        if (first_read + RB) % 4 == 0 and g + 2 < NG:
            fetch(g + 2)
        for u, run in enumerate(rest):                  # the build slice's share of this gap
            if slot_of[u] == k:
                run()
        if k >= NSETS - 1:
            accumulate(k - (NSETS - 1))
    for k in range(max(0, NB - (NSETS - 1)), NB):
        accumulate(k)
    for u, run in enumerate(rest):
        if slot_of[u] == NB:
            run()
    valid_lines = st.lines
    # ---------------- skipped position: the build slice alone
    sk = Stream()
    for run in filler_units(sk, tiles, KS, 's', rm):
        run()
    text = ['s_cmp_eq_u32 %[ok], 0', 's_cbranch_scc1 .Lpskip%='] + valid_lines + ['s_branch .Lpend%=', '.Lpskip%=:'] + sk.lines
    body = "".join('               "%s\\n\\t"\n' % s for s in text) + '               ".Lpend%=:"\n'
    outs = ", ".join('[c%d] "+v"(acc[%d])' % (i, i) for i in range(2 * reads))
    ins = ['[blk] "v"(blk)', '[b] "v"(base)', '[ok] "s"(valid)', '[opA] "v"(opA)', '[opB] "v"(opB)', '[m0] "s"(mA)', '[m1] "s"(mB)']
    ins += ['[i%d] "n"(IMM0 + %d)' % (g, 8 * g) for g in range(NG)]
    ins += ['[oa%d] "n"(OPO + %d)' % (ks, ks * 1024) for ks in range(KS)]
    ins += ['[ob%d_%d] "n"(OPO + %d)' % (t, ks, ks * 2048 + t * 64) for t in range(2) for ks in range(KS)]
    clob = ", ".join('"v%d"' % r for r in range(rm.first, 256))
    return ('// CPW %d, %d positions, position %d, %d k-step(s): %d look-ups + result tiles %s of the wave; fixed registers v[%d:255]\n'
            'template <int IMM0, int OPO>\n'
            '__device__ __forceinline__ void gpi_%d_%d_%d(f32x2* acc, uint32_t blk, uint32_t base, int valid, uint32_t opA, uint32_t opB,\n'
            '                                            uint32_t mA, uint32_t mB) {\n'
            '  asm volatile(\n%s'
            '               : %s\n'
            '               : %s\n'
            '               : "scc", "m0", "memory", %s);\n}\n'
            % (CPW, NP, P, KS, 2 * reads, tiles, rm.first, CPW, KS, P, body, outs, ", ".join(ins), clob))


def main():
    print('// GENERATED by scripts/gen/gen_sym8i.py — do not edit; see that script for the schedule and the register map.')
    print('#ifndef QCNN_SYM8I_POS_H_\n#define QCNN_SYM8I_POS_H_\n')
    for CPW, NP in ((48, 2), (32, 3), (24, 4), (16, 6)):
        for KS in (1, 2):
            for P in range(NP):
                print(emit(CPW, NP, P, KS))
    print('#endif  // QCNN_SYM8I_POS_H_')


if __name__ == "__main__":
    main()
