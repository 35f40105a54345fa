#!/usr/bin/env python3
"""Generate quantized-cnn_amd/csrc/qcnn_sym8_gather.h: the look-ups of ONE output position of k_conv_sym8 as one hand-scheduled
asm statement per position, software-pipelined over blocks of RB ds_read_b128.

Why a generator: an asm statement needs literal text, and the schedule — which counted s_waitcnt goes where — depends on the
number of reads per position (CPW / 2 = 8, 12, 16, 24), the reads per block RB and the depth of the pipeline.  With NSETS sets
of read temporaries at the top of the 256-register file the reads of blocks k + 1 .. k + NSETS - 1 are in flight while block k
is accumulated (eight waves per CU cannot hide an LDS round trip per block behind each other the way sixteen could):

    F(0) F(1)                         F(g): ds_read_b64 of the four 16-bit row offsets of read group g from the program row
    for block k:   [wait F(group of k)]  X(k) R(k)  [F(g + 2) once group g is consumed]  A(k - NSETS + 1)
    drain:         A(...) of the last NSETS - 1 blocks

X: RB v_xor_b32_sdwa (offset + lane address), R: RB ds_read_b128, A: 2 RB v_pk_add_f32.  LDS operations return in order, so
every wait is a COUNT of the LDS operations issued after the one waited for — computed here by simulating the issue order.
usage: python scripts/gen/gen_sym8_gather.py [RB NSETS] > quantized-cnn_amd/csrc/qcnn_sym8_gather.h
"""
import sys


def emit(reads, RB, NSETS):
    """asm text + operand lists for a position of `reads` ds_read_b128 (= 2 * reads look-ups per wave)."""
    NB = reads // RB                     # blocks
    NG = reads // 4                      # offset groups (one ds_read_b64 = four 16-bit offsets = four reads)
    top = 256
    set_base = [top - (NSETS - s) * RB * 4 for s in range(NSETS)]
    slot_base = set_base[0] - 4          # two offset slots of two registers
    slots = [(slot_base, slot_base + 1), (slot_base + 2, slot_base + 3)]
    L, issued = [], []                   # asm lines; LDS operations in issue order (names)
    a = L.append

    def wait_for(name):
        n = len(issued) - 1 - issued.index(name)
        a('s_waitcnt lgkmcnt(%d)' % n)

    def fetch(g):
        lo, hi = slots[g & 1]
        a('ds_read_b64 v[%d:%d], %%[blk] offset:%%[i%d]' % (lo, hi, g))
        issued.append('F%d' % g)

    def accumulate(k):
        base = set_base[k % NSETS]
        for r in range(RB):
            wait_for('R%d_%d' % (k, r))
            c = 2 * (k * RB + r)
            a('v_pk_add_f32 %%[c%d], v[%d:%d], %%[c%d]' % (c, base + 4 * r, base + 4 * r + 1, c))
            a('v_pk_add_f32 %%[c%d], v[%d:%d], %%[c%d]' % (c + 1, base + 4 * r + 2, base + 4 * r + 3, c + 1))

    a('s_cmp_eq_u32 %[ok], 0')
    a('s_cbranch_scc1 .Lpskip%=')
    fetch(0)
    if NG > 1:
        fetch(1)
    for k in range(NB):
        base = set_base[k % NSETS]
        first_read = k * RB
        g = first_read // 4
        if first_read % 4 == 0:
            wait_for('F%d' % g)
        for r in range(RB):
            q = (first_read + r) % 4                     # which of the group's four offsets
            src = slots[g & 1][q >> 1]
            a('v_xor_b32_sdwa v%d, v%d, %%[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
              % (base + 4 * r, src, q & 1))
        for r in range(RB):
            a('ds_read_b128 v[%d:%d], v%d' % (base + 4 * r, base + 4 * r + 3, base + 4 * r))
            issued.append('R%d_%d' % (k, r))
        if (first_read + RB) % 4 == 0 and g + 2 < NG:     # group g consumed: its slot takes group g + 2
            fetch(g + 2)
        if k >= NSETS - 1:
            accumulate(k - (NSETS - 1))
    for k in range(max(0, NB - (NSETS - 1)), NB):
        accumulate(k)
    text = "".join('               "%s\\n\\t"\n' % s for s in L) + '               ".Lpskip%=:"\n'
    outs = ", ".join('[c%d] "+v"(acc[%d])' % (i, i) for i in range(2 * reads))
    imms = ", ".join('[i%d] "n"(IMM0 + %d)' % (g, 8 * g) for g in range(NG))
    clob = ", ".join('"v%d"' % r for r in range(slot_base, 256))
    return ('// %d reads = %d look-ups of one position (blocks of %d reads, %d sets of temporaries v[%d:255], offset slots v[%d:%d]);\n'
            '// acc[2j], acc[2j+1] = the four images of channel j of the lane half\n'
            'template <int IMM0>\n'
            '__device__ __forceinline__ void gpos%d(f32x2* acc, uint32_t blk, uint32_t base, int valid) {\n'
            '  asm volatile(\n%s'
            '               : %s\n'
            '               : [blk] "v"(blk), [b] "v"(base), [ok] "s"(valid), %s\n'
            '               : "scc", %s);\n}\n' % (reads, 2 * reads, RB, NSETS, set_base[0], slot_base, slot_base + 3, reads // 4,
                                                   text, outs, imms, clob))


def emit_f16(reads, RB, NSETS):
    """The same position statement over an fp16 table (QCNN_OPT_LUT_MODE = 2, BASELINE configs[4]): a row is 256 bytes, a lane's
    four images of a row are ONE ds_read_b64 (half the LDS cycles of the ds_read_b128 of the f32 table), and every half is
    added to its fp32 sum by one v_fma_mix_f32 (sum + half * 1.0, rounded once: exactly the fp32 addition of the converted
    value) — four per read where the f32 table takes two v_pk_add_f32.  Same pipeline: RB reads per block, NSETS sets of
    two-register temporaries, counted waits from the simulated LDS queue."""
    NB = reads // RB
    NG = reads // 4
    top = 256
    set_base = [top - (NSETS - s) * RB * 2 for s in range(NSETS)]
    slot_base = set_base[0] - 4
    slots = [(slot_base, slot_base + 1), (slot_base + 2, slot_base + 3)]
    L, issued = [], []
    a = L.append

    def wait_for(name):
        n = len(issued) - 1 - issued.index(name)
        a('s_waitcnt lgkmcnt(%d)' % n)

    def fetch(g):
        lo, hi = slots[g & 1]
        a('ds_read_b64 v[%d:%d], %%[blk] offset:%%[i%d]' % (lo, hi, g))
        issued.append('F%d' % g)

    def accumulate(k):
        base = set_base[k % NSETS]
        for r in range(RB):
            wait_for('R%d_%d' % (k, r))
            c = 4 * (k * RB + r)
            for e in range(4):
                a('v_fma_mix_f32 %%[c%d], v%d, 1.0, %%[c%d] op_sel:[%d,0,0] op_sel_hi:[1,0,0]' % (c + e, base + 2 * r + (e >> 1), c + e, e & 1))

    a('s_cmp_eq_u32 %[ok], 0')
    a('s_cbranch_scc1 .Lhskip%=')
    fetch(0)
    if NG > 1:
        fetch(1)
    for k in range(NB):
        base = set_base[k % NSETS]
        first_read = k * RB
        g = first_read // 4
        if first_read % 4 == 0:
            wait_for('F%d' % g)
        for r in range(RB):
            q = (first_read + r) % 4
            src = slots[g & 1][q >> 1]
            a('v_xor_b32_sdwa v%d, v%d, %%[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
              % (base + 2 * r, src, q & 1))
        for r in range(RB):
            a('ds_read_b64 v[%d:%d], v%d' % (base + 2 * r, base + 2 * r + 1, base + 2 * r))
            issued.append('R%d_%d' % (k, r))
        if (first_read + RB) % 4 == 0 and g + 2 < NG:
            fetch(g + 2)
        if k >= NSETS - 1:
            accumulate(k - (NSETS - 1))
    for k in range(max(0, NB - (NSETS - 1)), NB):
        accumulate(k)
    text = "".join('               "%s\\n\\t"\n' % s for s in L) + '               ".Lhskip%=:"\n'
    outs = ", ".join('[c%d] "+v"(acc[%d].%s)' % (i, i // 2, "xy"[i & 1]) for i in range(4 * reads))
    imms = ", ".join('[i%d] "n"(IMM0 + %d)' % (g, 8 * g) for g in range(NG))
    clob = ", ".join('"v%d"' % r for r in range(slot_base, 256))
    return ('// fp16 table: %d reads = %d look-ups of one position (blocks of %d ds_read_b64, %d sets of temporaries v[%d:255], offset\n'
            '// slots v[%d:%d]); acc[2j], acc[2j+1] = the four images of channel j of the lane half (fp32 sums)\n'
            'template <int IMM0>\n'
            '__device__ __forceinline__ void hpos%d(f32x2* acc, uint32_t blk, uint32_t base, int valid) {\n'
            '  asm volatile(\n%s'
            '               : %s\n'
            '               : [blk] "v"(blk), [b] "v"(base), [ok] "s"(valid), %s\n'
            '               : "scc", %s);\n}\n' % (reads, 2 * reads, RB, NSETS, set_base[0], slot_base, slot_base + 3, reads // 4,
                                                   text, outs, imms, clob))


def emit_f16acc(reads, RB, NSETS):
    """fp16 table AND fp16 running sums (QCNN_OPT_LUT_MODE = 3, the second column of the configs[4] study): the four images of a
    channel are two packed registers, a read is followed by two v_pk_add_f16 (round to nearest even after every addition, like
    the oracle's qo_study_mode(1, 1)).  Half the accumulator registers of the fp32 sums: the kernel gives a wave twice the
    (position, channel) pairs."""
    NB = reads // RB
    NG = reads // 4
    top = 256
    set_base = [top - (NSETS - s) * RB * 2 for s in range(NSETS)]
    slot_base = set_base[0] - 4
    slots = [(slot_base, slot_base + 1), (slot_base + 2, slot_base + 3)]
    L, issued = [], []
    a = L.append

    def wait_for(name):
        n = len(issued) - 1 - issued.index(name)
        a('s_waitcnt lgkmcnt(%d)' % n)

    def fetch(g):
        lo, hi = slots[g & 1]
        a('ds_read_b64 v[%d:%d], %%[blk] offset:%%[i%d]' % (lo, hi, g))
        issued.append('F%d' % g)

    def accumulate(k):
        base = set_base[k % NSETS]
        for r in range(RB):
            wait_for('R%d_%d' % (k, r))
            c = 2 * (k * RB + r)
            a('v_pk_add_f16 %%[c%d], %%[c%d], v%d' % (c, c, base + 2 * r))
            a('v_pk_add_f16 %%[c%d], %%[c%d], v%d' % (c + 1, c + 1, base + 2 * r + 1))

    a('s_cmp_eq_u32 %[ok], 0')
    a('s_cbranch_scc1 .Laskip%=')
    fetch(0)
    if NG > 1:
        fetch(1)
    for k in range(NB):
        base = set_base[k % NSETS]
        first_read = k * RB
        g = first_read // 4
        if first_read % 4 == 0:
            wait_for('F%d' % g)
        for r in range(RB):
            q = (first_read + r) % 4
            src = slots[g & 1][q >> 1]
            a('v_xor_b32_sdwa v%d, v%d, %%[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
              % (base + 2 * r, src, q & 1))
        for r in range(RB):
            a('ds_read_b64 v[%d:%d], v%d' % (base + 2 * r, base + 2 * r + 1, base + 2 * r))
            issued.append('R%d_%d' % (k, r))
        if (first_read + RB) % 4 == 0 and g + 2 < NG:
            fetch(g + 2)
        if k >= NSETS - 1:
            accumulate(k - (NSETS - 1))
    for k in range(max(0, NB - (NSETS - 1)), NB):
        accumulate(k)
    text = "".join('               "%s\\n\\t"\n' % s for s in L) + '               ".Laskip%=:"\n'
    outs = ", ".join('[c%d] "+v"(acc[%d])' % (i, i) for i in range(2 * reads))
    imms = ", ".join('[i%d] "n"(IMM0 + %d)' % (g, 8 * g) for g in range(NG))
    clob = ", ".join('"v%d"' % r for r in range(slot_base, 256))
    return ('// fp16 table, fp16 sums: %d reads = %d look-ups of one position (blocks of %d ds_read_b64, %d sets of temporaries v[%d:255],\n'
            '// offset slots v[%d:%d]); acc[2j] = images 0, 1 and acc[2j+1] = images 2, 3 of channel j of the lane half, packed fp16\n'
            'template <int IMM0>\n'
            '__device__ __forceinline__ void apos%d(uint32_t* acc, uint32_t blk, uint32_t base, int valid) {\n'
            '  asm volatile(\n%s'
            '               : %s\n'
            '               : [blk] "v"(blk), [b] "v"(base), [ok] "s"(valid), %s\n'
            '               : "scc", %s);\n}\n' % (reads, 2 * reads, RB, NSETS, set_base[0], slot_base, slot_base + 3, reads // 4,
                                                   text, outs, imms, clob))


def main():
    RB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    NSETS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    print('// GENERATED by scripts/gen/gen_sym8_gather.py %d %d — do not edit; see that script for the schedule.' % (RB, NSETS))
    print('#ifndef QCNN_SYM8_GATHER_H_\n#define QCNN_SYM8_GATHER_H_\n')
    for reads in (8, 12, 16, 24):        # CPW = 16, 24, 32, 48 -> gpos2, gpos3, gpos4, gpos6 (named by CPW / 8)
        print(emit(reads, RB, NSETS))
    for reads in (8, 12, 16, 24):        # the same positions over an fp16 table -> hpos2, hpos3, hpos4, hpos6
        print(emit_f16(reads, RB, NSETS))
    for reads in (8, 12, 16, 24):        # ... with fp16 running sums -> apos2, apos3, apos4, apos6
        print(emit_f16acc(reads, RB, NSETS))
    print('#endif  // QCNN_SYM8_GATHER_H_')


if __name__ == "__main__":
    main()
