#!/usr/bin/env python
"""Static check for the one hazard the compiler cannot see in the hand-scheduled kernels: an instruction that touches a register
whose `ds_read` (issued from inline asm, completing asynchronously) has not been waited for.  The compiler believes an asm
output is defined AT the asm statement, so it is free to copy it (phi resolution on a loop exit, tuple assembly) before the
source's own `s_waitcnt lgkmcnt`: the copy then carries the register's OLD content.  (Round 3: exactly that on the exit edge of
attn96.hip's loops -- one launch in ~15 at 24 heads had one 32x32 block of one item off; tools/probes/race96.py.)

Linear scan of a kernel in layout order: LDS reads queue their destination registers; `s_waitcnt lgkmcnt(N)` retires all but the N
youngest (LDS reads return in order); any other instruction naming a queued register is reported.
usage: python tools/audit_async_lds.py file.s kernel_substring"""
import re
import sys


def regs_of(text):
    out = set()
    for kind, a, b in re.findall(r'\b([va])\[(\d+):(\d+)\]', text):
        out.update((kind, i) for i in range(int(a), int(b) + 1))
    for kind, a in re.findall(r'\b([va])(\d+)\b', text):
        out.add((kind, int(a)))
    return out


def audit(lines, start, end, max_states=200000):
    """Path-sensitive over the kernel's control-flow graph: a state = (basic block, the in-order queue of LDS reads in flight)."""
    # ---- basic blocks
    label_at = {}
    instrs = []                                   # (line, text)
    for i in range(start, end):
        raw = lines[i].split(';')[0].rstrip()
        t = raw.strip()
        if not t or t.startswith('.') and not t.endswith(':'):
            continue
        if t.endswith(':'):
            label_at[t[:-1]] = len(instrs)
            continue
        instrs.append((i, t))
    leaders = {0} | set(label_at.values())
    for k, (_, t) in enumerate(instrs):
        if t.split()[0].startswith(('s_branch', 's_cbranch', 's_endpgm', 's_setpc')):
            leaders.add(k + 1)
    leaders = sorted(x for x in leaders if x < len(instrs))
    block_of = {}
    for bi, l in enumerate(leaders):
        block_of[l] = bi
    ends = leaders[1:] + [len(instrs)]

    def successors(bi):
        last = instrs[ends[bi] - 1][1]
        op = last.split()[0]
        out = []
        if op.startswith('s_endpgm') or op.startswith('s_setpc'):
            return out
        if op.startswith(('s_branch', 's_cbranch')):
            tgt = last.split()[-1]
            if tgt in label_at and label_at[tgt] in block_of:
                out.append(block_of[label_at[tgt]])
            if op.startswith('s_branch'):
                return out
        if ends[bi] < len(instrs):
            out.append(block_of[ends[bi]])
        return out

    bad = {}
    seen = set()
    work = [(0, ())]
    while work:
        bi, pend = work.pop()
        if (bi, pend) in seen:
            continue
        seen.add((bi, pend))
        if len(seen) > max_states:
            raise RuntimeError("state explosion")
        pending = list(pend)                      # [(line, frozenset(regs))] oldest first
        for k in range(leaders[bi], ends[bi]):
            i, t = instrs[k]
            op = t.split()[0]
            if op.startswith('s_waitcnt'):
                m = re.search(r'lgkmcnt\((\d+)\)', t)
                if m:
                    n = int(m.group(1))
                    pending = pending[len(pending) - n:] if n else []
                continue
            if op.startswith(('ds_read', 'ds_load')):
                body = t[len(op):]
                dest, rest = body.split(',', 1)
                used = regs_of(rest)
                for p in pending:
                    if p[1] & used:
                        bad.setdefault(i, (t, p[0]))
                pending.append((i, frozenset(regs_of(dest))))
                pending = pending[-15:]           # (the counter holds 15; older reads have returned by the time a 16th issues)
                continue
            if op.startswith('s_'):
                continue
            if op.startswith('ds_'):                  # LDS writes / atomics without a return value: they count, they define nothing
                used = regs_of(t[len(op):])
                for p in pending:
                    if p[1] & used:
                        bad.setdefault(i, (t, p[0]))
                pending.append((i, frozenset()))
                pending = pending[-15:]
                continue
            used = regs_of(t)
            for p in pending:
                if p[1] & used:
                    bad.setdefault(i, (t, p[0]))
        for sb in successors(bi):
            work.append((sb, tuple(pending)))
    return [(i, t, src) for i, (t, src) in sorted(bad.items())]


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l)] + [len(lines)]
    total = 0
    for a, b in zip(starts[:-1], starts[1:]):
        if sub not in lines[a]:
            continue
        bad = audit(lines, a, b)
        total += len(bad)
        print(lines[a].split(':')[0], '->', len(bad), 'uses of registers with an LDS read in flight')
        for i, t, src in bad[:12]:
            print(f'   line {i + 1}: {t[:90]}    (read issued at line {src + 1})')
    return total


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
