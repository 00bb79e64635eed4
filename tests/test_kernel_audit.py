"""Static audit of the one-wave-per-SIMD attention kernels (attn64.hip, attn96.hip).  They address the accumulator half of the
register file BY NAME from inline asm and list every accumulator register as clobbered; the moment hipcc runs out of VGPRs it
uses accumulator registers as spill space regardless -- silent corruption of O^T.  The audit after every edit:
compile to assembly for gfx950 (no GPU needed), then per kernel: no VGPR spills, no scratch, and no `v_accvgpr_*` instruction
outside the `#ASMSTART ... #ASMEND` regions the source wrote itself.  Also: the generated slot tables are up to date."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chipmunk_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asm(tmp_path, name):
    out = tmp_path / (name + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           os.path.join(CSRC, name + ".hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    return out.read_text()


@pytest.mark.parametrize("name,kernels", [("attn96", ["csp96_kernel"]), ("attn64", ["attn64_kernel", "colsum64_kernel"])])
def test_named_accumulator_kernels_have_no_spills_and_no_compiler_accvgpr(tmp_path, name, kernels):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = _asm(tmp_path, name)
    # ---- metadata: spills / scratch per kernel
    meta = re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, flags=re.S)
    if not meta:   # field order differs between compiler versions: fall back to per-field scans
        names = re.findall(r"^\s+\.name:\s+(\S+)", text, flags=re.M)
        spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
        scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)
        meta = list(zip(names, scratch, spills))
    checked = 0
    for kname, scratch, spill in meta:
        if any(k in kname for k in kernels):
            assert int(spill) == 0, f"{kname}: {spill} VGPR spills (they land in the accumulator registers that hold O^T)"
            assert int(scratch) == 0, f"{kname}: {scratch} bytes of scratch"
            checked += 1
    assert checked >= len(kernels), (checked, [m[0] for m in meta])
    # ---- no accumulator moves the compiler made up
    func, inasm, bad = None, False, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            func = m.group(1)
        if "#ASMSTART" in line:
            inasm = True
        elif "#ASMEND" in line:
            inasm = False
        elif "v_accvgpr" in line and not inasm and func and any(k in func for k in kernels):
            bad.append((func, line.strip()))
    assert not bad, f"compiler-generated accumulator moves: {bad[:4]} ... ({len(bad)} in all)"
    # ---- no instruction touches a register whose (asynchronous, inline-asm) LDS read has not been waited for: the compiler takes
    #      an asm output as defined AT the asm and may copy it -- phi resolution on a loop exit, tuple assembly -- before the
    #      source's own s_waitcnt (round 3: exactly that on the exit edge of attn96's loops, one launch in ~15 slightly wrong);
    #      path-sensitive over the kernel's CFG, LDS reads retire in order (tools/audit_async_lds.py); this also checks every
    #      hand-counted lgkmcnt of the kernels
    import importlib.util
    spec = importlib.util.spec_from_file_location("audit_async_lds", os.path.join(ROOT, "tools", "audit_async_lds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l)] + [len(lines)]
    audited = 0
    for a, b in zip(starts[:-1], starts[1:]):
        if any(k in lines[a] for k in kernels):
            hits = mod.audit(lines, a, b)
            assert not hits, f"{lines[a].split(':')[0]}: registers used while their LDS read is in flight: {hits[:4]}"
            audited += 1
    assert audited >= len(kernels)


def test_attn96_slot_tables_are_current(tmp_path):
    """attn96_sched.h is generated (tools/gen_attn96_sched.py asserts the ordering constraints of the schedule): the committed
    header must be what the generator emits now."""
    hdr = os.path.join(CSRC, "attn96_sched.h")
    before = open(hdr).read()
    try:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_attn96_sched.py")], stdout=subprocess.DEVNULL)
        assert open(hdr).read() == before
    finally:
        open(hdr, "w").write(before)


def test_topk_mask_kernels_do_not_spill(tmp_path):
    """topk_mask_kernel keeps a row's keys in registers (60 packed VGPRs at 119 056 columns, 1024 threads = 128 VGPRs per thread) and
    writes the mask from them in an unrolled loop: one hoisted unpacking of the keys and it spills 89 registers (seen).  Every
    instantiation must compile without spills or scratch."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = _asm(tmp_path, "indexed_io")
    names = re.findall(r"\.name:\s+(\S+)", text)
    scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
    assert len(names) == len(scratch) == len(spills)
    seen = 0
    for kname, sc, sp in zip(names, scratch, spills):
        if "topk_mask_kernel" in kname:
            assert int(sp) == 0 and int(sc) == 0, f"{kname}: {sp} spills, {sc} bytes of scratch"
            seen += 1
    assert seen >= 6


def _asm_statements(text):
    """(template, outputs) of every inline-asm statement: the body split at ':' OUTSIDE string literals."""
    for m in re.finditer(r"\basm\s+volatile\s*\(|\basm\s*\(", text):
        depth, i, instr = 0, m.end() - 1, False
        while i < len(text):            # the statement's parenthesised body (parentheses inside strings do not count)
            ch = text[i]
            if instr:
                if ch == "\\":
                    i += 1
                elif ch == '"':
                    instr = False
            elif ch == '"':
                instr = True
            elif ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        body = text[m.end():i]
        parts, cur, instr, j = [], [], False, 0
        while j < len(body):
            ch = body[j]
            if instr:
                cur.append(ch)
                if ch == "\\":
                    cur.append(body[j + 1])
                    j += 1
                elif ch == '"':
                    instr = False
            elif ch == '"':
                instr = True
                cur.append(ch)
            elif ch == ":":
                parts.append("".join(cur))
                cur = []
            else:
                cur.append(ch)
            j += 1
        parts.append("".join(cur))
        template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
        template = re.sub(r"\\[nt]", " ", template)     # the source spells line breaks as backslash-n / backslash-t
        yield template, (parts[1] if len(parts) > 1 else "")


def test_multi_read_asm_statements_have_early_clobber_outputs():
    """An inline-asm statement that holds TWO OR MORE memory reads writing registers must declare its outputs early-clobber
    ("=&v"): without it the compiler may give a result the register of a (dying) address operand, and when the second read queues
    behind the first one, the first one's return overwrites the address before the second has issued.  Seen in GEMM2's fragment
    reads (round 4): intermittent wrong 32 x 32 blocks, never the same place twice, green on most runs.  Source-level scan of every
    kernel file (no compiler needed)."""
    bad, seen = [], 0
    for fn in sorted(os.listdir(CSRC)):
        if not fn.endswith((".hip", ".h")):
            continue
        for template, outputs in _asm_statements(open(os.path.join(CSRC, fn)).read()):
            reads = len(re.findall(r"\b(?:ds_read|ds_load|buffer_load|global_load|s_load)\w*", template))
            regs = re.findall(r'"(=[^"]*)"', outputs)
            if reads < 2 or not regs:
                continue
            seen += 1
            if any("&" not in r for r in regs):
                bad.append(f"{fn}: {template[:70]}... outputs {regs}")
    assert seen, "the scan found no multi-read asm statement at all: its parser is broken"
    assert not bad, "multi-read asm without early-clobber outputs:\n" + "\n".join(bad)


def _vgprs(operand):
    """Set of VGPR numbers an operand like v7 or v[4:7] names (accumulator / scalar operands: empty)."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("name,kernels", [("attn96", ["csp96_kernel"]), ("attn64", ["attn64_kernel", "colsum64_kernel"])])
def test_no_valu_write_to_an_mfma_source_right_in_front_of_an_asm_mfma(tmp_path, name, kernels):
    """the kernels issue their MFMAs as inline asm, which the compiler's hazard recogniser cannot see into.  If register
    allocation ever makes it assemble an A / B operand with VALU copies (v_mov, v_perm, v_cndmask ...) directly in front of such a
    statement, no wait states separate the VALU write from the matrix core's read.  The drain MFMAs carry an explicit `s_nop 4`; the
    main loops rely on their operands being LDS-read results or long-lived registers.  This walks the compiled kernels and fails if any
    VGPR source of an asm MFMA is written by a VALU instruction within the 4 instructions in front of it (s_nop counted as its
    wait states)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    text = _asm(tmp_path, name)
    bad, checked = [], 0
    for kname in re.findall(r"^(\S+):\s*(?:;.*)?$", text, flags=re.M):
        if not any(k in kname for k in kernels) or not kname.startswith("_Z"):
            continue
        body = text[text.index("\n" + kname + ":"):]
        body = body[:body.index(".Lfunc_end")]
        insts = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith((";", ".", "//")) and not ln.strip().endswith(":")]
        for i, ins in enumerate(insts):
            if not ins.startswith("v_mfma"):
                continue
            ops = [o.strip() for o in ins.split(None, 1)[1].split(",")]
            srcs = _vgprs(ops[1]) | _vgprs(ops[2])
            if not srcs:
                continue
            checked += 1
            slack = 0
            for prev in reversed(insts[max(0, i - 6):i]):
                if prev.startswith("s_nop"):
                    slack += int(prev.split()[1]) + 1
                    continue
                if slack >= 4:
                    break
                slack += 1
                if prev.startswith("v_") and not prev.startswith(("v_mfma", "v_accvgpr_read")):
                    dst = prev.split(None, 1)[1].split(",")[0].strip()
                    if _vgprs(dst) & srcs:
                        bad.append(f"{kname[:40]}: `{prev}` writes a source of `{ins[:60]}`")
    assert checked > 20, "no asm MFMA with VGPR sources found: the walk is broken"
    assert not bad, "VALU write to an MFMA source without wait states:\n" + "\n".join(bad[:10])
