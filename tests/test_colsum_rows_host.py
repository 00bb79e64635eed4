"""Host check of `colsum_part_rows` (csrc/attn_params.h): the mapping from a 192-row query group to the one or two partial column-sum rows the
one-pass mask step writes per 256-row workgroup (attn64.hip MODE 3, in-workgroup combine) -- the mask kernel and cs_combine read through it, the
attention kernel writes by its own arithmetic (first nA = 3 - (4g mod 3) waves of workgroup g form its first set).  Compiled for the host with
hipcc and compared with a brute-force model: every 64-row wave block must be covered exactly once by the rows of its group."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

SRC = r'''
#include <cstdio>
#include "attn_params.h"
int main() {
    for (int nq : {64, 192, 256, 300, 1100, 4352, 32760, 119056}) {
        const int nwg = (nq + 255) / 256, groups = (nq + 191) / 192;
        for (int j = 0; j < groups; ++j) {
            int r0, r1;
            const int n = colsum_part_rows(j, nwg, r0, r1);
            printf("%d %d %d %d %d\n", nq, j, n, r0, n == 2 ? r1 : -1);
        }
    }
    return 0;
}
'''


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_colsum_part_rows_cover_every_wave_block_once(tmp_path):
    src = tmp_path / "rows.cpp"
    src.write_text(SRC)
    exe = tmp_path / "rows"
    subprocess.check_call([HIPCC, "-x", "hip", "--offload-arch=gfx950", "-O1", "-I", os.path.join(ROOT, "chipmunk_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True)
    got = {}
    for line in out.split("\n"):
        if line.strip():
            nq, j, n, r0, r1 = map(int, line.split())
            got.setdefault(nq, {})[j] = [r0] + ([r1] if n == 2 else [])
    for nq, rows in got.items():
        nwg = (nq + 255) // 256
        # what the kernel writes: workgroup g's waves 0 .. nA-1 -> row 2g (group of wave block 4g), waves nA .. 3 -> row 2g + 1
        row_of_block, group_of_row = {}, {}
        for g in range(nwg):
            n_a = 3 - (4 * g) % 3
            for w in range(4):
                wb = 4 * g + w
                row = 2 * g + (0 if w < n_a else 1)
                row_of_block[wb] = row
                group_of_row.setdefault(row, set()).add(wb // 3)
        for row, gs in group_of_row.items():
            assert len(gs) == 1, f"Nq {nq}: partial row {row} mixes groups {gs}"
        for j, rs in rows.items():
            blocks = [wb for wb in (3 * j, 3 * j + 1, 3 * j + 2) if wb < 4 * nwg]
            want = sorted({row_of_block[wb] for wb in blocks})
            assert sorted(rs) == want, f"Nq {nq} group {j}: reads rows {rs}, its wave blocks {blocks} were written to {want}"
