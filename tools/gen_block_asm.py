#!/usr/bin/env python3
"""Generates poppunk_amd/csrc/ppk_block_asm.inc: the hand-scheduled gfx950 instruction
stream for ONE 64-bin block (14 bit-planes) of the 4-ref x 4-query register tile.

Why assembly: on MI355X a v_bitop3_b32 whose src0 and src1 sit in the same VGPR bank
(register number mod 4) issues ~1.5x slower (tools/ubench_bank.hip: 4.7 vs 3.1 clk).  hipcc
allocates the 32 accumulators and 16 operand registers without regard to banks, so the
stream is emitted with fixed registers instead:

  operands   ds_read_b128 puts a sample's (lo, hi) dwords in (even, odd) registers, so
             every "lo" compare reads banks {0,2} and every "hi" compare banks {1,3};
  accumulators  lo accumulators live in ODD registers, hi accumulators in EVEN registers:
             no v_bitop3 ever has src0 in the bank of src1 or src2.

The loads of plane b+1 are issued under the compares of plane b (query words are
double-buffered in registers, ref words are re-loaded as soon as their 16 compares are out).

Register map (all clobbered by the block):
  v[ACC .. ACC+31]  accumulators: pair p = 4*r + q -> hi = ACC + 2p, lo = ACC + 2p + 1
  v[A0 .. A0+3]     refs 2l, 2l+1      (x,y | z,w)        v[A1 .. A1+3]  refs 128+2l, 128+2l+1
  v[S0 .. S0+7]     query words, even planes            v[S1 .. S1+7]  odd planes
"""
import os
import sys

BB = 14


class Map:
    """Fixed-register map for a 4-ref x TQ-query tile (all bases multiples of 4: bank = component)."""

    def __init__(self, tq):
        self.TQ = tq
        self.A0, self.A1 = 72, 76
        self.B0, self.B1 = 48, 52      # second ref-operand buffer (GEN_A2: odd planes)
        self.S0 = 80
        self.S1 = self.S0 + 2 * tq
        self.ACC = self.S1 + 2 * tq
        self.END = self.ACC + 8 * tq       # one past the last clobbered register

    def lo_acc(self, r, q):
        return self.ACC + 2 * (self.TQ * r + q) + 1

    def hi_acc(self, r, q):
        return self.ACC + 2 * (self.TQ * r + q)

    def a_reg(self, r, hi, plane=0, a2=False):      # r in 0..3 -> register of ref r's lo/hi dword
        if a2 and (plane & 1):
            base = self.B0 if r < 2 else self.B1
        else:
            base = self.A0 if r < 2 else self.A1
        return base + 2 * (r & 1) + (1 if hi else 0)

    def s_reg(self, plane, q, hi):
        base = self.S0 if (plane & 1) == 0 else self.S1
        return base + 2 * q + (1 if hi else 0)


REF_PLANE_BYTES = 2048      # 256 samples x 8 B
REF_HALF_BYTES = 1024


def gen(QRY_PLANE_BYTES, TQ=4, dma_planes=None, half=False, a2=False):
    """TQ = 4: counters %[c0]..%[c15], one per pair (p = 4r + q).
    TQ = 8: counters %[c0]..%[c15], two pairs per counter (pair p = 8r + q -> counter p >> 1,
    16-bit half p & 1; a block adds at most 64 and a k at most 16 * 64 per pair ... callers
    must drain the counters before a half can reach 65536)."""
    m = Map(TQ)
    A0, A1, S0, S1 = m.A0, m.A1, m.S0, m.S1
    lo_acc, hi_acc, a_reg, s_reg = m.lo_acc, m.hi_acc, m.a_reg, m.s_reg
    NS = TQ // 2                 # ds_read_b128 per query plane
    out = []
    emit = out.append

    def dma(t):
        """LDS-DMA piece t of the NEXT block, issued from inside the compare stream (a VALU-only
        stretch: the piece costs the issuing wave far fewer cycles there than next to the block's
        opening burst of ds_reads).  Operands: %[m00] / %[m03] LDS byte address of piece 0 / 3
        (pieces 1, 2 are 8 KB apart), %[sb0..3] 64-bit global base of each piece, %[voa] / %[vob]
        per-lane byte offset of pieces 0-2 / 3, %[xblo],%[xbhi] exec mask of piece 3 (the last query
        piece has fewer rows).  After the last block the caller lets the pieces re-load that block."""
        if t == 0:
            emit("s_mov_b32 m0, %[m00]")
        elif t < 3:
            emit("s_add_u32 m0, %%[m00], %d" % (8192 * t))
        else:
            emit("s_mov_b32 m0, %[m03]")
        if t == 3:      # (a 64-bit "s" operand is not reliably kept in SGPRs by the compiler: halves)
            emit("s_mov_b32 exec_lo, %[xblo]")
            emit("s_mov_b32 exec_hi, %[xbhi]")
        else:
            emit("s_nop 0")     # m0 write -> LDS-DMA needs one wait state
        emit("global_load_lds_dwordx4 %%[%s], %%[sb%d]" % ("voa" if t < 3 else "vob", t))
        if t == 3:
            emit("s_mov_b64 exec, -1")

    def load_s(plane):
        base = S0 if (plane & 1) == 0 else S1
        for i in range(NS):
            emit("ds_read_b128 v[%d:%d], %%[qp] offset:%d"
                 % (base + 4 * i, base + 4 * i + 3, plane * QRY_PLANE_BYTES + 16 * i))

    def load_a0(plane):
        base = m.B0 if (a2 and (plane & 1)) else A0
        emit("ds_read_b128 v[%d:%d], %%[rp] offset:%d" % (base, base + 3, plane * REF_PLANE_BYTES))

    def load_a1(plane):
        base = m.B1 if (a2 and (plane & 1)) else A1
        emit("ds_read_b128 v[%d:%d], %%[rp] offset:%d" % (base, base + 3, plane * REF_PLANE_BYTES + REF_HALF_BYTES))

    def ops(plane, rs):
        for r in rs:
            if dma_planes and r == 3 and plane in dma_planes:
                dma(dma_planes.index(plane))
            for q in range(TQ):
                for hi in (0, 1):
                    acc = hi_acc(r, q) if hi else lo_acc(r, q)
                    a, s = a_reg(r, hi, plane, a2), s_reg(plane, q, hi)
                    if plane == 0:
                        emit("v_xnor_b32 v%d, v%d, v%d" % (acc, a, s))
                    else:
                        emit("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x90" % (acc, acc, a, s))

    # Wave priority falls through the block: 3 at plane 0, 2 at plane 4, 1 at plane 8, 0 at plane 12.  Every
    # block ends in the workgroup's barrier, which waits for the SLOWEST of its 8 wavefronts; of two
    # wavefronts on a SIMD the one that is earlier in its block is the one a barrier is waiting for, so it
    # gets the issue slots first.  Measured on MI355X (tools/ab_so.py, same box): -1.8..-2.5 % kernel time;
    # the reverse schedule (0,1,2,3) +4.7 %; two levels or a later fall less.  (GEN_PRIO=a,b,c,d or one value
    # per plane overrides; GEN_ALIGN=<log2> aligns the block's first instruction: no effect measured.)
    if os.environ.get("GEN_ALIGN") and dma_planes:
        emit(".p2align %d" % int(os.environ["GEN_ALIGN"]))
    prio = None
    if dma_planes:       # (the full block of the packed modes; on the half block it measured 1 % worse)
        v = [int(x) for x in os.environ.get("GEN_PRIO", "3,2,1,0").split(",")]
        prio_cnt = v.pop() if len(v) == BB + 1 else None      # (a 15th value: the popcount tail)
        prio = v if len(v) == BB else [v[b // 4] for b in range(BB)]
    if half:
        # Diagonal tiles whose queries all lie beyond the tile's first 128 refs: refs 0/1 of every
        # lane pair with nothing, so only refs 2/3 (the a1 operands) are compared: half the stream.
        load_s(0)
        load_a1(0)
        for b in range(BB):
            last = b == BB - 1
            if prio and (b == 0 or prio[b] != prio[b - 1]):
                emit("s_setprio %d" % prio[b])
            if not last:
                load_s(b + 1)
                emit("s_waitcnt lgkmcnt(%d)" % NS)     # s(b), a1(b) landed; s(b+1) may be pending
            else:
                emit("s_waitcnt lgkmcnt(0)")
            ops(b, (2, 3))
            if not last:
                load_a1(b + 1)
        for r in (2, 3):
            for q in range(TQ):
                p = TQ * r + q
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p, lo_acc(r, q), p))
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p, hi_acc(r, q), p))
        return out
    if a2:
        # Ref operands double-buffered like the query operands: ALL four reads of plane b+1 are issued
        # before the 32 compares of plane b, so every operand has a full plane of the wave's own issue
        # time to arrive (single-buffered: half a plane), and one s_waitcnt per plane instead of two.
        # A wavefront then stalls less when fewer of its SIMD's other wavefronts are there to cover for
        # it -- while the other workgroup is at a barrier or in its epilogue.  (GEN_A2=1; measured
        # 1 % SLOWER than the single-buffered schedule, same box: not used.)
        load_s(0)
        load_a0(0)
        load_a1(0)
        for b in range(BB):
            last = b == BB - 1
            if prio and (b == 0 or prio[b] != prio[b - 1]):
                emit("s_setprio %d" % prio[b])
            if not last:
                load_s(b + 1)
                load_a0(b + 1)
                load_a1(b + 1)
                emit("s_waitcnt lgkmcnt(%d)" % (NS + 2))   # plane b's operands have landed
            else:
                emit("s_waitcnt lgkmcnt(0)")
            ops(b, (0, 1, 2, 3))
        for r in range(4):
            for q in range(TQ):
                p = TQ * r + q
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p, lo_acc(r, q), p))
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p, hi_acc(r, q), p))
        return out
    # prologue: s(0), a0(0), a1(0)
    load_s(0)
    load_a0(0)
    load_a1(0)
    for b in range(BB):
        last = b == BB - 1
        if prio and (b == 0 or prio[b] != prio[b - 1]):
            emit("s_setprio %d" % prio[b])
        if not last:
            load_s(b + 1)
            emit("s_waitcnt lgkmcnt(%d)" % (NS + 1))   # s(b) and a0(b) have landed; a1(b), s(b+1) may be pending
        else:
            emit("s_waitcnt lgkmcnt(1)")      # only a1(b) may be pending
        ops(b, (0, 1))
        if not last:
            load_a0(b + 1)
            emit("s_waitcnt lgkmcnt(%d)" % (NS + 1))   # a1(b) landed; s(b+1), a0(b+1) may be pending
        else:
            emit("s_waitcnt lgkmcnt(0)")
        ops(b, (2, 3))
        if not last:
            load_a1(b + 1)
    # popcount-accumulate into the compiler-visible counters %[c0] .. %[c15]
    if dma_planes and prio_cnt is not None and prio_cnt != prio[-1]:
        emit("s_setprio %d" % prio_cnt)
    for r in range(4):
        for q in range(TQ):
            p = TQ * r + q
            if TQ == 4:
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p, lo_acc(r, q), p))
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p, hi_acc(r, q), p))
            elif (p & 1) == 0:
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p >> 1, lo_acc(r, q), p >> 1))
                emit("v_bcnt_u32_b32 %%[c%d], v%d, %%[c%d]" % (p >> 1, hi_acc(r, q), p >> 1))
            else:
                t = A0 + (p >> 1) % 8        # operand registers are dead by now: scratch
                emit("v_bcnt_u32_b32 v%d, v%d, 0" % (t, lo_acc(r, q)))
                emit("v_bcnt_u32_b32 v%d, v%d, v%d" % (t, hi_acc(r, q), t))
                emit("v_lshl_add_u32 %%[c%d], v%d, 16, %%[c%d]" % (p >> 1, t, p >> 1))
    return out


def write_macro(f, name, lines):
    f.write("#define %s \\\n" % name)
    for ln in lines:
        f.write('  "%s\\n" \\\n' % ln)
    f.write("  \"\"\n")
    print("wrote", name, len(lines), "instructions")


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    # product: what dist_kernel_v2 (256 x 32 tile) runs
    dst = os.path.join(os.path.dirname(here), "poppunk_amd", "csrc", "ppk_block_asm.inc")
    # experiments (tools/ubench_pipe.hip only): the rejected 256 x 64 tile and 4x8 register tile
    dst_x = os.path.join(here, "ppk_block_asm_experiments.inc")
    m4, m8 = Map(4), Map(8)
    a2 = os.environ.get("GEN_A2", "0") == "1"
    clob = ", ".join('"v%d"' % i for i in (list(range(m4.B0, m4.B0 + 8)) if a2 else []) + list(range(m4.A0, m4.END)))
    clob8 = ", ".join('"v%d"' % i for i in range(m8.A0, m8.END))
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_block_asm.py -- do not edit.  One 64-bin block (14 planes) of the\n"
                "// 4x4 register tile with bank-aware fixed VGPRs v%d..v%d; see the generator for the map.\n"
                "// _Q32: 32 queries per workgroup tile (query plane stride 256 bytes).\n"
                % (m4.A0, m4.END - 1))
        write_macro(f, "PPK_BLOCK_ASM_Q32", gen(256, a2=a2))
        f.write("// the same with the 4 LDS-DMA pieces of the next block issued inside the stream\n")
        # (GEN_DMA_PLANES=a,b,c,d: experiments with the placement of the four pieces)
        planes = [int(x) for x in os.environ.get("GEN_DMA_PLANES", "1,4,7,10").split(",")]
        write_macro(f, "PPK_BLOCK_DMA_ASM_Q32", gen(256, 4, dma_planes=planes, a2=a2))
        f.write("// refs 2/3 only (diagonal tiles with every query beyond the first 128 refs)\n")
        write_macro(f, "PPK_BLOCK_HALF_ASM_Q32", gen(256, 4, half=True))
        f.write("#define PPK_BLOCK_ASM PPK_BLOCK_ASM_Q32\n")
        f.write("#define PPK_BLOCK_CLOBBERS %s\n" % clob)
    if "--experiments" not in sys.argv[1:]:
        return
    # measured-and-rejected shapes: generated on demand, not tracked (tools/ubench_pipe.hip needs them)
    with open(dst_x, "w") as f:
        f.write("// GENERATED by tools/gen_block_asm.py -- do not edit.  Measured-and-rejected shapes, used by\n"
                "// tools/ubench_pipe.hip only (not part of libppk_hip.so): _Q64 = 64 queries per workgroup tile\n"
                "// (16 wavefronts), BLOCK8 = 4x8 register tile (v%d..v%d: 16 counters, two 16-bit counts each).\n"
                % (m8.A0, m8.END - 1))
        write_macro(f, "PPK_BLOCK_ASM_Q64", gen(512))
        for name, stride in (("PPK_BLOCK8_ASM_Q32", 256), ("PPK_BLOCK8_ASM_Q64", 512)):
            write_macro(f, name, gen(stride, 8))
        f.write("#define PPK_BLOCK8_CLOBBERS %s\n" % clob8)


if __name__ == "__main__":
    main()
