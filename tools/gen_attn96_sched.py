#!/usr/bin/env python
"""Slot schedules of attn96.hip's main loop -> chipmunk_amd/csrc/attn96_sched.h (constexpr tables) -- with every ordering
constraint the hand-written version kept in comments CHECKED here.

One 32-key tile = 48 MFMA slots: even slot s = QK^T (query block s/16, k step (s%16)/2), odd slot s = 2i+1 = PV element i.
Two PV orders:

* PAIR (the kernel's loop without a reference point, NOMAX): elements of an iteration t
    i = 0..2    block 2 of tile t-2, V^T fragments f = 5, 6, 7
    i = 3..18   PAIRS of tile t-1: fragment f = (i-3)/2 for query blocks 0 and 1 back to back -- one fragment read, two MFMAs
    i = 19..23  block 2 of tile t-1, fragments f = 0..4
  so a tile reads 16 V^T fragments (32 ds_read_b64_tr_b16) instead of 24 (48).  P[0] is double-buffered by tile parity (its
  packs for tile t start at slot 30, its reads for tile t-1 end at slot 35).  Not usable with a moving reference point: the
  rescale of block 0's accumulators (slot 23) would fall between PV elements of that block.
* MAJOR (running-maximum loop): query-block-major as before -- i < 3: block 2 of t-2 (f = 5..7), then blocks 0, 1 (8 each) and
  five elements of block 2 of tile t-1; every block's elements precede the slot of that block's reference update.

LDS reads are spread one or two per slot over ALL slots (the hand-written schedule put 2-3 after every PV MFMA and none after
the QK^T ones: 6-8 issues in odd slots against 3 in even ones, and a slot cannot be shorter than its MFMA).
lgkmcnt values: LDS reads return in order; the count in front of a consumer = reads issued after the last one it needs.
"""
import os

NS = 48
UPDATE_SLOT = {2: 7, 0: 23, 1: 39}        # kernel: update_block(qb) (reference move + rescale of the block's accumulators)


def pack_slot(qb, K):
    """(iteration offset, slot) at which pair K (slab K/4) of block qb of tile t is written: the kernel's window W of block qb
    runs at slot SG with qb 2: W = SG-8 (next iteration); qb 1: W = SG-40 (SG >= 40), W = SG+8 (next iteration); qb 0: W = SG-24;
    pair K is packed at W = 2K+6."""
    W = 2 * K + 6
    if qb == 2:
        return (1, W + 8)
    if qb == 0:
        return (0, W + 24)
    return (0, W + 40) if W + 40 < NS else (1, W - 8)


def build(mode):
    elems = []                                   # (query block, fragment f = slab*4 + db, use index u)
    for i in range(24):
        if i < 3:
            elems.append((2, 5 + i, i))
        elif mode == "pair":
            if i < 19:
                e = i - 3
                elems.append((e % 2, e // 2, 3 + e // 2))
            else:
                elems.append((2, i - 19, 11 + (i - 19)))
        else:
            elems.append(((i - 3) // 8 if i < 19 else 2, (i - 3) % 8 if i < 19 else i - 19, i))
    n_uses = max(e[2] for e in elems) + 1
    assert n_uses % 4 == 0
    use_f, use_c = {}, {}
    for i, (qb, f, u) in enumerate(elems):
        use_f[u] = f
        use_c.setdefault(u, []).append(i)
    # ---- LDS reads per slot: ('V0'|'V1', use) halves of a fragment, ('Q', ks) Q^T block-1 window, ('K', ks) K(t+1) fragment
    reads = [[] for _ in range(NS)]
    if mode == "pair":
        vstart = {3: 0, 4: 4, 5: 8, 6: 12, 7: 16, 8: 20, 9: 24, 10: 28, 11: 26, 12: 30, 13: 34, 14: 38, 15: 40, 0: 42, 1: 44, 2: 46}
    else:
        vstart = {u: (2 * u - 6) % NS for u in range(24)}
    for u, s0 in vstart.items():
        reads[s0].append(('V0', u))
        reads[s0 + 1].append(('V1', u))
    qslot = {0: 2, 1: 6, 2: 10, 3: 14, 4: 18, 5: 19, 6: 22, 7: 23}
    for ks, s in qslot.items():
        reads[s].append(('Q', ks))
    for ks in range(8):
        reads[33 + 2 * ks].append(('K', ks))
    assert max(len(r) for r in reads) <= 2, [len(r) for r in reads]

    # ---- P[qb][slab] of the element's tile: complete before the element, not overwritten by the next tile's packs before it
    for i, (qb, f, u) in enumerate(elems):
        slab = f // 4
        lag = 2 if i < 3 else 1                  # tile t - lag, computed in iteration t - lag
        Ks = range(slab * 4, slab * 4 + 4)
        ready = max((-lag + pack_slot(qb, K)[0]) * NS + pack_slot(qb, K)[1] for K in Ks)
        assert ready < 2 * i + 1, (mode, i, qb, f, ready)
        step = 2 if qb == 0 else 1               # P[0] is double-buffered by tile parity
        over = min((-lag + step + pack_slot(qb, K)[0]) * NS + pack_slot(qb, K)[1] for K in Ks)
        assert over > 2 * i + 1, (mode, i, qb, f, over)
        if mode == "major":                      # a block's elements of tile t-1 precede its reference update for tile t
            upd = UPDATE_SLOT[qb] + (NS if qb == 2 and i >= 3 else 0)   # block 2 of tile t-1 is updated at slot 7 of iteration t+1
            assert 2 * i + 1 + 2 <= upd, (i, qb, upd)
    # ---- V window (4 entries, entry = u & 3).  Use cycle c: uses 0..2 are READ in iteration c-1 (slots 42..47) and consumed in
    #      iteration c; the others are read and consumed in iteration c.  A use's halves are issued only after the last consumer of
    #      the entry's previous occupant (use u-4; u < 4: use u-4+n_uses of the previous cycle), >= 5 slots before its first consumer.
    def abs_read(u, c):
        return ((c - 1) if u < 3 else c) * NS + vstart[u]
    def abs_cons(u, c):
        return [c * NS + 2 * i + 1 for i in use_c[u]]
    for u in range(n_uses):
        pu, pc = (u - 4, 0) if u >= 4 else (u - 4 + n_uses, -1)
        assert max(abs_cons(pu, pc)) <= abs_read(u, 0), (mode, u, pu)
        assert min(abs_cons(u, 0)) - (abs_read(u, 0) + 1) >= 5, (mode, u, "fragment read too late")
        assert vstart[u] + 1 < NS                # both halves in one iteration: the fragment's LDS tile is V(t-1) of the READING iteration
    # Q window (entry ks & 3): k step ks+4 overwrites k step ks after block 1's MFMA on ks (slot 16 + 2 ks)
    for ks in range(8):
        if ks >= 4:
            assert qslot[ks] >= 16 + 2 * (ks - 4)
        assert 16 + 2 * ks - qslot[ks] >= 5
    # K(t+1) fragment ks overwrites K(t)'s after block 2's MFMA on k step ks (slot 32 + 2 ks): read at slot 33 + 2 ks

    # ---- lgkmcnt: program order = per slot: [wait] MFMA, then the slot's reads; three iterations back to back
    seq = [(it * NS + s, kind, arg) for it in range(-1, 2) for s in range(NS) for kind, arg in reads[s]]
    per_iter = len(seq) // 3
    def need(s):
        if s % 2 == 0:
            qb, ks = s // 16, (s % 16) // 2
            return [('K', ks, -1)] if qb == 0 else [('Q', ks, 0)] if qb == 1 else []
        qb, f, u = elems[(s - 1) // 2]
        off = -1 if u < 3 else 0
        return [('V0', u, off), ('V1', u, off)]
    start = -10 ** 9
    for _ in range(4):                           # iterate to the steady state of "what has landed at the top of an iteration"
        landed = start
        waits = [-1] * NS
        for s in range(NS):
            idx_needed = -1
            for kind, arg, off in need(s):
                c = [n for n, (a, k, g) in enumerate(seq) if k == kind and g == arg and (a // NS) == off]
                assert len(c) == 1 and seq[c[0]][0] < s, (mode, s, kind, arg)
                idx_needed = max(idx_needed, c[0])
            issued = max([n for n, (a, k, g) in enumerate(seq) if a < s], default=-1)
            if idx_needed > landed:
                waits[s] = issued - idx_needed
                landed = idx_needed
        start = landed - per_iter
    assert all(-1 <= w <= 15 for w in waits)

    KIND = {'V0': 1, 'V1': 2, 'Q': 3, 'K': 4}
    def arr(name, vals):
        return f"    static constexpr int {name}[{len(vals)}] = {{" + ", ".join(str(v) for v in vals) + "};\n"
    out = f"struct {'Pair' if mode == 'pair' else 'Major'} {{\n"
    out += arr("PV_QB", [e[0] for e in elems]) + arr("PV_F", [e[1] for e in elems]) + arr("PV_U", [e[2] & 3 for e in elems])
    for j in range(2):
        out += arr(f"RD_KIND{j}", [KIND[r[j][0]] if len(r) > j else 0 for r in reads])
        out += arr(f"RD_ARG{j}", [((r[j][1] & 3) * 8 + use_f[r[j][1]] if r[j][0][0] == 'V' else r[j][1]) if len(r) > j else 0 for r in reads])
    out += arr("WAIT", waits)
    # the drain after the loop: the three tail elements (uses 0..2 read in the last iteration)
    out += "};\n"
    return out, [len(r) for r in reads]


text = ("// generated by tools/gen_attn96_sched.py (the ordering constraints are asserted there) -- do not edit\n#pragma once\n"
        "// per PV element i (odd slot 2i+1): query block, V^T fragment (slab*4 + d block), window entry; per slot: up to two LDS reads\n"
        "// after the MFMA (kind 1/2 = half 0/1 of a fragment, arg = entry*8 + fragment; 3 = Q^T block-1 window k step; 4 = K(t+1) k step);\n"
        "// WAIT = lgkmcnt in front of the slot's MFMA (-1: none)\nnamespace a96s {\n")
for mode in ("pair", "major"):
    t, n = build(mode)
    text += t
    print(mode, "LDS reads per slot:", n)
text += "}  // namespace a96s\n"
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chipmunk_amd", "csrc", "attn96_sched.h")
open(path, "w").write(text)
print(text)
