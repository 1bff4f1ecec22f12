#!/usr/bin/env python3
"""Generates crag_q16_kloop.inc for crag_scan_q16.hip: the per-tile statements of the query-stationary MFMA scan
(v_mfma_f32_16x16x32_bf16).

Output tile per wave: 32 corpus rows x 64 queries (256 queries per workgroup of four waves) = 2 row blocks x 4 query
blocks of 16x16 accumulators (8 x 4 registers).  Every A fragment read from LDS feeds four MFMAs (one per query
block).  The 32x32x16 form of rounds 1-2 (2 query blocks of 32x32, same tile, LDS image, LDS reads, DMA, registers and
flops) ran at a lower clock on random data -- MI355X_MICROARCH.md ("DVFS give-back", item 7) -- and was retired after
the same-box A/B of round 3.

REGISTER FILE PARTITION (one wave per SIMD: 256 VGPRs + 256 AGPRs).  At D = 768 the stationary query fragments alone
are 96 x 4 = 384 registers; hipcc's allocator, given them as asm operands, spills them wholesale (two uses each: they
look cold).  They are therefore kept OUT of the compiler's sight, in fixed registers that only the generated
statements name; every statement lists them as clobbers, so the compiler holds nothing there ACROSS a statement, and
build.py (check_q16_isa) proves on the emitted ISA that it uses none of them BETWEEN statements either:
  a[0:255], v[128:255]  query fragments: fragment n = qb * (KS / 2) + s (query block qb = 0..3, 32-element k-step s)
                        in a[4n:4n+3] for n < 64, else v[128 + 4 (n - 64) : +3]   [fixed registers, ISA-checked]
  v[96:127]             accumulators: (row block rb, query block qb) -> v[96 + 16 rb + 4 qb : +3]; pinned OUTPUTS
  v[68:75]              xs[4 rb + e] = |x|^2 of row 16 rb + 4 (lane >> 4) + e: the C operand of the tile's FIRST MFMAs (clobber)
  v[76:95]              the five rotating A fragments (clobbers)
  v[64:67]              the four XORed fragment addresses of the tile (clobbers)
  v[0:63]               the compiler's
THE SCORE IS THE ACCUMULATOR.  The scan's score is s = |x|^2 - 2 q.x (l2) or -q.x (ip).  The B operand holds the
queries PRE-SCALED by -2 (l2) or -1 (ip) -- exact in bf16; k_q16_scale_queries writes the copy once per search -- and
the first MFMA of every accumulator takes C = |x|^2 (read from the tile's |x|^2 slot in LDS; a zeroed slot for ip)
instead of 0.  The accumulator then IS the score: no epilogue arithmetic, the filter's fast path is a v_min3 tree over
the lane's 8 rows per query block (16 VALU instructions per tile where the 32x32x16 statement spends 48 + 16), and the
slow path and the sample store read the accumulators as they are.  The certificate's error bound counts the extra term
(crag_select.hip: every partial sum is bounded by |x|^2 + 2 |q||x| instead of 2 |q||x|).
Lane map (cdna_hip_programming.md section 3): A fragment = row (lane & 15) of the row block, k = 8 (lane >> 4) + [0, 8)
of the 32-element k-step; B fragment = query (lane & 15) of the query block, same k; accumulator register e = row
4 (lane >> 4) + e of the row block, column = query (lane & 15).  So the QUERY stays on the lane (tau is a VGPR per
query block) and a query's 32 rows of a tile sit on four lanes (lane >> 4 = 0..3), eight rows each.

TWO FORMS of the statement.  Q16KLoop (plain) is self-contained: it reads its own operands from the start and touches
nothing outside itself; the kernel's general path (sample launches, the first and last tiles of a stream, ragged
tiles, waves without queries) uses it with the DMA issued up front by compiler code.  Q16Fast is the steady state of a
filter launch: it carries the LDS-DMA of the tile two steps ahead between its MFMAs, ends with the end-of-tile
synchronisation (counted vmcnt, s_barrier) and then ISSUES THE FIRST READS OF THE NEXT TILE (|x|^2 runs, the four XORed
addresses, fragments 0..3) before its own min tree, so their LDS latency hides behind the min tree, the hit test and the
loop bookkeeping instead of stalling the next statement's first MFMAs.  Between two Q16Fast statements v[64:95] are
therefore LIVE (addresses, |x|^2, fragments 0..3); the compiler never allocates that high (it needs fewer than 60
registers) and build.py proves on the emitted ISA that no compiler instruction of the kernel touches v[64:95].
Q16First issues the same first reads for the tile that opens a run of Q16Fast statements.

A "half-step" h = 2 s + rb multiplies row block rb of k-step s into the four query blocks: four MFMAs of 16 cycles,
64 cycles of matrix pipe.  LDS address of (s, rb): (a0 ^ 64 (s & 3)) + 256 (s >> 2) + rb * 16 rows:
the chunk swizzle (4 s + g) ^ (row & 15) folds into one XOR because rows are multiples of 256 bytes, and rows r and
r + 16 swizzle alike, so both row blocks share the XORed address and differ in the immediate offset.

    python gen_q16_kloop.py > crag_q16_kloop.inc
"""
import os
DIST = int(os.environ.get("Q16_DIST", "3"))  # half-steps an LDS read runs ahead of its MFMAs (3 or 4)
NT, WAVES = DIST + 2, 4
PRE = DIST + 1  # fragments read ahead of a tile's first MFMA (by the previous statement in the steady state)
SPREAD = os.environ.get("Q16_ORDER", "spread") != "clump"
ABL = set(filter(None, os.environ.get("Q16_ABL", "").split(",")))  # (comma-separated) ablation builds for tools/q16_bench.hip only: nodma | noepi | nomfma | nolds | nobar
# rotating A fragments: five in v[76:95]; a sixth (DIST = 4) takes over v[68:71], free once the first MFMAs have read their C
T = [f"v[{76 + 4 * i}:{79 + 4 * i}]" for i in range(5)] + ["v[68:71]"] * (NT - 5)
assert NT <= 6
ADDRS = ["v64", "v65", "v66", "v67"]  # a0 ^ 64 j: the chunk swizzle of k-steps s = j (mod 4)
XS0 = 68
def acc0(rb, qb):
    return 96 + 16 * rb + 4 * qb
def acc(rb, qb):
    return f"v[{acc0(rb, qb)}:{acc0(rb, qb) + 3}]"
def frag(KS, qb, s):
    n = qb * (KS // 2) + s
    return f"a[{4 * n}:{4 * n + 3}]" if n < 64 else f"v[{128 + 4 * (n - 64)}:{131 + 4 * (n - 64)}]"
def frag_clobbers(KS):
    n = 2 * KS
    regs = [f'"a{i}"' for i in range(4 * min(n, 64))] + [f'"v{128 + i}"' for i in range(4 * max(0, n - 64))]
    for r in ('"a255"', '"v255"'):  # the highest registers of both files always: the kernel is allocated all 512
        if r not in regs:
            regs.append(r)
    return ", ".join(regs)
print("// GENERATED by gen_q16_kloop.py -- do not edit.")
def first_reads(L, lds_ops, a0, xa, ROWB):
    """the |x|^2 runs (C operand of the first MFMAs), the four XORed fragment addresses, fragments 0..3"""
    for r in range(2):
        lds_ops.append(("X", r))
        L.append(f"ds_read_b128 v[{XS0 + 4 * r}:{XS0 + 3 + 4 * r}], {xa} offset:{64 * r}")
    L.append(f"v_mov_b32 {ADDRS[0]}, {a0}")
    for j in range(1, 4):
        L.append(f"v_xor_b32 {ADDRS[j]}, {64 * j}, {a0}")
    for h in range(PRE):
        s, rb = h >> 1, h & 1
        lds_ops.append(("A", h))
        L.append(f"ds_read_b128 {T[h % NT]}, {ADDRS[s & 3]} offset:{(s >> 2) * 256 + rb * 16 * ROWB}")
def min_tree(L, mn):
    for qb in range(4):
        a = acc0(0, qb)
        L.append(f"v_min3_f32 {mn[qb]}, v{a}, v{a + 1}, v{a + 2}")
    for qb in range(4):
        a, b = acc0(0, qb), acc0(1, qb)
        L.append(f"v_min3_f32 {mn[qb]}, {mn[qb]}, v{a + 3}, v{b}")
    for qb in range(4):
        b = acc0(1, qb)
        L.append(f"v_min3_f32 {mn[qb]}, {mn[qb]}, v{b + 1}, v{b + 2}")
    for qb in range(4):
        L.append(f"v_min_f32 {mn[qb]}, {mn[qb]}, v{acc0(1, qb) + 3}")
TCLOB = ", ".join('"v%d"' % i for i in range(64, 96))
print("template <int KS> struct Q16KLoop;")
print("template <int KS> struct Q16First;")
print("template <int KS, int PH, bool NT> struct Q16Fast;")
for KS in (16, 24, 32, 48):  # KS = row length / 16 = half-steps per tile
    PPW = KS // WAVES
    ROWB = KS * 32
    assert KS >= 8
    # ---- plain statement --------------------------------------------------------------------------------------
    ops = []
    def op(c, e):
        ops.append((c, e))
        return f"%{len(ops) - 1}"
    op('=&{v[96:111]}', 'acc0')   # row block 0: element 4 qb + e
    op('=&{v[112:127]}', 'acc1')  # row block 1
    mn = [op('=&v', f'mn{j}') for j in range(4)]
    n_out = len(ops)
    a0 = op('v', 'a0')
    xa = op('v', 'xaddr')
    L = ["s_waitcnt lgkmcnt(0)"]
    lds_ops = []  # LDS reads in program order; LDS returns in order
    first_reads(L, lds_ops, a0, xa, ROWB)
    def rd(h):
        s, rb = h >> 1, h & 1
        lds_ops.append(("A", h))
        return [f"ds_read_b128 {T[h % NT]}, {ADDRS[s & 3]} offset:{(s >> 2) * 256 + rb * 16 * ROWB}"]
    def wait_for(tag):
        return f"s_waitcnt lgkmcnt({len(lds_ops) - 1 - lds_ops.index(tag)})"
    def mfmas(h):
        s, rb = h >> 1, h & 1
        out = []
        for qb in range(4):
            c = f"v[{XS0 + 4 * rb}:{XS0 + 4 * rb + 3}]" if s == 0 else acc(rb, qb)
            out.append(f"v_mfma_f32_16x16x32_bf16 {acc(rb, qb)}, {T[h % NT]}, {frag(KS, qb, s)}, {c}")
        return out
    # One wave per SIMD: an MFMA holds the vector issue for 8 of its 16 cycles, so each MFMA has ONE short shadow; the
    # half-step's other instructions (the wait, the LDS read three half-steps ahead, a DMA piece, its M0) go one per
    # shadow instead of in a clump behind the fourth MFMA (SPREAD; Q16_ORDER=clump restores the clump for A/B builds)
    for h in range(KS):
        m = mfmas(h)
        if SPREAD:
            L.append(wait_for(("A", h)))  # the X reads are older than every A read: landed as well
            L.append(m[0])
            if PRE <= h + DIST < KS:
                L += rd(h + DIST)
            L += m[1:]
        else:
            if PRE <= h + DIST < KS:
                L += rd(h + DIST)
            L.append(wait_for(("A", h)))
            L += m
    # the filter's fast path: mn[qb] = min over the lane's 8 rows of the accumulators (= the scores).  hipcc inserts no
    # hazard nops inside an asm statement; a 4-pass MFMA result needs 7 wait states (11 if it counts as 8 passes) before
    # a VALU instruction reads it: 12 behind the last MFMA
    L.append("s_nop 11")
    min_tree(L, mn)
    body = "\\n\\t".join(L)
    outs = ", ".join(f'"{c}"({e})' for c, e in ops[:n_out])
    ins = ", ".join(f'"{c}"({e})' for c, e in ops[n_out:])
    print(f"template <> struct Q16KLoop<{KS}> {{")
    print("    static __device__ __forceinline__ void run(f32x16_t& acc0, f32x16_t& acc1, float& mn0, float& mn1, float& mn2, float& mn3, unsigned a0, unsigned xaddr) {")
    print(f'        asm volatile("{body}"')
    print(f"                     : {outs}")
    print(f"                     : {ins}")
    print(f"                     : \"memory\", {TCLOB}, {frag_clobbers(KS)});")
    print("    }")
    print("};")
    # ---- the first reads of the tile that opens a run of Q16Fast statements ---------------------------------------
    L = ["s_waitcnt lgkmcnt(0)"]
    first_reads(L, [], "%0", "%1", ROWB)
    body = "\\n\\t".join(L)
    print(f"template <> struct Q16First<{KS}> {{")
    print("    static __device__ __forceinline__ void run(unsigned a0, unsigned xaddr) {")
    print(f'        asm volatile("{body}" : : "v"(a0), "v"(xaddr) : "memory", {TCLOB});')
    print("    }")
    print("};")
    # ---- steady-state statement -----------------------------------------------------------------------------------
    for NONTEMP in (False, True):
        for PH in range(4):
            ntbit = " nt" if NONTEMP else ""
            ops = []
            op('=&{v[96:111]}', 'acc0')
            op('=&{v[112:127]}', 'acc1')
            mn = [op('=&v', f'mn{j}') for j in range(4)]
            keep = op('=&s', 'keep')
            POLL = PH == 0 and not NONTEMP  # wave 0 of a launch with several query groups: pacing poll
            pv = op('=&s', 'pv') if POLL else None
            n_out = len(ops)
            a0n = op('v', 'a0n')
            xan = op('v', 'xaddrn')
            voff = [op('v', f'd.voff[{i}]') for i in range(PPW // 2)]
            base = [op('s', 'd.tile'), op('s', 'd.tile_hi')]
            ldst = op('s', 'd.lds_dst')
            xoff = op('v', 'd.xoff')
            xbase = op('s', 'd.xn')
            xdst = op('s', 'd.xn_dst')
            pp = op('s', 'pp') if POLL else None
            # fragments 0..3, the |x|^2 runs and the addresses of THIS tile were issued by the previous statement
            L = ["s_waitcnt lgkmcnt(0)"]
            if POLL:  # the progress words of the workgroups that read the same tiles: one s_load here, landed long
                L.append(f"s_load_dwordx8 {pv}, {pp}, 0x0 glc")  # before the tail's lgkmcnt(0); SMEM shares lgkmcnt with the
            lds_ops = [("S", 0)] if POLL else []                  # LDS reads and returns out of order: counted as the oldest,
            L.append(f"s_mov_b32 {keep}, m0")                     # it only makes the counted waits stricter while in flight
            slots = []  # (half-step, M0 of the piece, the load): M0 is set right behind the previous piece
            for h in range(KS):
                if h % 4 == PH and "nodma" not in ABL:
                    i, hh = (h // 4) % (PPW // 2), (h // 4) // (PPW // 2)  # piece i of tile half hh (rows 16 hh ..)
                    slots.append((h, f"s_add_u32 m0, {ldst}, {(i * WAVES + hh * (KS // 2)) * 1024}", f"global_load_lds_dwordx4 {voff[i]}, {base[hh]}{ntbit}"))
                if h == 4 + (PH + 2) % 4 and "nodma" not in ABL:
                    slots.append((h, f"s_mov_b32 m0, {xdst}", f"global_load_lds_dword {xoff}, {xbase}{ntbit}"))
            assert len(slots) == PPW + 1 or "nodma" in ABL
            if slots:
                L.append(slots[0][1])
            nslot = 0
            for h in range(KS):
                m = mfmas(h) if "nomfma" not in ABL else [None] * 4
                here = []
                while nslot < len(slots) and slots[nslot][0] == h:
                    here.append(slots[nslot][2])
                    nslot += 1
                    here.append(slots[nslot][1] if nslot < len(slots) else f"s_mov_b32 m0, {keep}")
                assert len(here) <= 2
                if SPREAD:
                    if ("A", h) in lds_ops:
                        L.append(wait_for(("A", h)))
                    L.append(m[0])
                    if PRE <= h + DIST < KS:
                        L += rd(h + DIST)
                    L.append(m[1])
                    L += here[:1]   # the DMA piece
                    L.append(m[2])
                    L += here[1:]   # M0 of the next piece
                    L.append(m[3])
                else:
                    if PRE <= h + DIST < KS:
                        L += rd(h + DIST)
                    if ("A", h) in lds_ops:
                        L.append(wait_for(("A", h)))
                    L += m + here
                while L and L[-1] is None:
                    L.pop()
                L[:] = [x for x in L if x is not None]
            # end of tile: every wave retires all but the DMA instructions of THIS statement (its youngest; tile t + step
            # has then landed in its view), the barrier makes that true for all four, and the next tile's first reads go out
            L.append(f"s_waitcnt vmcnt({len(slots)}) lgkmcnt(0)")
            if "nobar" not in ABL:  # (timing ablation only: racy)
                L.append("s_barrier")
            first_reads(L, [], a0n, xan, ROWB)
            # (the wait, the barrier and these ten instructions stand between the last MFMA and the VALU reads of its result)
            if "noepi" in ABL:
                L += [f"v_mov_b32 {m}, 0x7f800000" for m in mn]
            else:
                L.append("s_nop 1")
                min_tree(L, mn)
            if "nolds" in ABL:  # (timing ablation only: the MFMAs multiply whatever the registers hold)
                L = [x for x in L if not x.startswith("ds_read")]
            body = "\\n\\t".join(L)
            outs = ", ".join(f'"{c}"({e})' for c, e in ops[:n_out])
            ins = ", ".join(f'"{c}"({e})' for c, e in ops[n_out:])
            extra = ", const unsigned* pp, u32x8_t& pv" if POLL else ""
            print(f"template <> struct Q16Fast<{KS}, {PH}, {'true' if NONTEMP else 'false'}> {{")
            print(f"    static __device__ __forceinline__ void run(f32x16_t& acc0, f32x16_t& acc1, float& mn0, float& mn1, float& mn2, float& mn3, unsigned a0n, unsigned xaddrn, const Q16Dma<{PPW}>& d{extra}) {{")
            print("        unsigned keep;")
            print(f'        asm volatile("{body}"')
            print(f"                     : {outs}")
            print(f"                     : {ins}")
            print(f"                     : \"memory\", {TCLOB}, {frag_clobbers(KS)});")
            print("    }")
            print("};")
# stationary query fragments: loaded from global memory straight into their fixed registers
print("template <int KS> struct Q16Load;")
for KS in (16, 24, 32, 48):
    print(f"template <> struct Q16Load<{KS}> {{")
    print("    static __device__ __forceinline__ void run(const unsigned short* p0, const unsigned short* p1, const unsigned short* p2, const unsigned short* p3) {")
    for qb in range(4):
        L = [f"global_load_dwordx4 {frag(KS, qb, s)}, %0, off offset:{64 * s}" for s in range(KS // 2)] + ["s_waitcnt vmcnt(0)"]
        body = "\\n\\t".join(L)
        print(f'        asm volatile("{body}" : : "v"(p{qb}) : "memory", {frag_clobbers(KS)});')
    print("    }")
    print("};")
