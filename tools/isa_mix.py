#!/usr/bin/env python3
"""Static VALU instruction mix of the compositing backward's hot loop (for bench.py's calibrated `roofline.valu_frac`).

Cross-compiles d3ga_amd/csrc/raster_composite_scan.hip to gfx950 assembly (no GPU needed), takes the group loop of
composite_bwd_tile_kernel<false, 512> -- the depth-2 loop (four pixels of one block line) counted four times, the rest of the
depth-1 loop once -- and classes every VALU instruction by the issue-cost classes tools/micro/valu_issue.hip measures:
plain (2-operand / fma), dpp, transcendental, packed, and other VOP3 (cndmask / cmp with an SGPR pair, med3, bfi ...).
Writes profiles/r05_composite_bwd_mix.json (one per round; bench.py reads the newest).
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "d3ga_amd", "csrc", "raster_composite_scan.hip")


def classify(op, line):
    if not op.startswith("v_"):
        return None
    if "_dpp" in op or "row_shr" in line or "quad_perm" in line or "row_ror" in line:
        return "dpp"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", op):
        return "trans"
    if op.startswith("v_pk_"):
        return "packed"
    if re.match(r"v_(cndmask_b32_e64|cmp\w*_e64|med3|bfi|mad_u64|lshl_add|add3|perm|readlane|writelane|readfirstlane)", op):
        return "vop3_other"
    return "plain"


def main():
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc",
                          "-fno-slp-vectorize", "-S", "--cuda-device-only", SRC, "-o", "-"], capture_output=True, text=True, check=True).stdout
    lines = asm.split("\n")
    kname = sys.argv[1] if len(sys.argv) > 1 else "_ZN4d3ga25composite_bwd_tile_kernelILb0ELi512"
    start = next(i for i, l in enumerate(lines) if l.startswith(kname))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the group loop = the depth-1 loop that contains the four-line loop (the kernel has small depth-1 loops in front of it)
    d1 = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and any("Child Loop" in x for x in body[i + 1:i + 6]))
    counts = {"plain": 0, "dpp": 0, "trans": 0, "packed": 0, "vop3_other": 0}
    other = {"lds": 0, "vmem": 0, "salu": 0}
    depth = 1
    first_d2 = True
    seen_d2 = 0
    for l in body[d1:]:
        if "Loop Header" in l and "Depth=2" in l:
            seen_d2 += 1
            first_d2 = seen_d2 == 1                          # the four-line pixel loop comes first; the insert loop is counted once per attempt
        m = re.search(r"Depth=(\d)", l)
        if m and ("Header" in l):
            depth = int(m.group(1))
        if "Depth=1" in l and "Loop Header" in l and l is not body[d1]:
            break                                            # the next top-level loop (final publish)
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if depth >= 3:
            continue                                         # the rare displaced-record publish inside the insert loop
        w = 4 if (depth == 2 and first_d2) else 1
        c = classify(op, t)
        if c:
            counts[c] += w
        elif op.startswith("ds_"):
            other["lds"] += w
        elif op.startswith("global_"):
            other["vmem"] += w
        elif op.startswith("s_"):
            other["salu"] += w
    out = {"kernel": kname, "unit": "instructions per 16-entry group (pixel-line loop x 4, one insert attempt)", "counts": counts,
           "non_valu": other, "note": "vop3_other is priced at the v_cndmask_b32_e64 / v_cmp_*_e64 / v_med3 / v_bfi rate measured by "
           "tools/micro/valu_issue.hip (1.9 ns vs 1.2 ns for v_fma_f32 at 8 waves/SIMD)", "vop3_other_cycles": 3.8}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_composite_bwd_mix.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
