#!/usr/bin/env python3
"""Compiles tcrc_engine.hip for gfx950 with --save-temps and counts, per kernel, the instructions of the basic
blocks that hold the row folds (those with at least 16 LDS lookups): row loads, LDS lookups, vector-ALU and
scalar instructions, and vector-ALU instructions per 16-byte row load and per lookup.  Needs no GPU.

    python tools/isa_counts.py > profiles/rNN_isa_counts.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "terarkdb_amd", "csrc", "tcrc_engine.hip")
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", SRC,
                           "-o", os.path.join(d, "e.o"), "--save-temps"], cwd=d, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, "tcrc_engine-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
want = {"k_crc_batchILi8ELb1ELi0E": "k_crc_batch<8 lanes, non-temporal> (headline kernel)",
        "k_crc_batchILi4ELb1ELi0E": "k_crc_batch<4 lanes, non-temporal>",
        "k_crc_batchILi8ELb1ELi1E": "k_crc_batch<8 lanes, non-temporal, segments> (long buffers)",
        "k_wal_fusedILb1E": "k_wal_fused<non-temporal> (log scan)"}
print("# gfx950 ISA of terarkdb_amd/csrc/tcrc_engine.hip: blocks with at least 16 ds_read_b32 (the row folds)")
print("# a row fold of one lane = one 16-byte load, 16 lookups (4 per dword): 4 x (4 v_perm_b32 + 4 v_xor_b32) = 32 VALU")
for i, l in enumerate(asm):
    m = re.match(r"(_Z\w+):", l)
    if not m or not any(k in m.group(1) for k in want):
        continue
    label = next(v for k, v in want.items() if k in m.group(1))
    end = next(k for k in range(i, len(asm)) if "s_endpgm" in asm[k])
    body = asm[i:end]
    total_valu = len([x for x in body if re.match(r"\s+v_", x)])
    n_insn = len([x for x in body if re.match(r"\s+[a-z]", x)])
    print(f"\n{label}: {n_insn} instructions, {total_valu} vector-ALU")
    blocks, cur = [], []
    for x in body:
        if re.match(r"\.LBB\d+_\d+:", x):
            blocks.append(cur)
            cur = []
        cur.append(x)
    blocks.append(cur)
    for b in blocks:
        t = "\n".join(b)
        ds = len(re.findall(r"ds_read_b32", t))
        if ds < 16:
            continue
        ld = len(re.findall(r"global_load_dwordx4", t))
        valu = len(re.findall(r"^\s+v_", t, re.M))
        perm = len(re.findall(r"v_perm_b32", t))
        salu = len(re.findall(r"^\s+s_", t, re.M))
        waits = sorted(set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", t)), key=int)
        name = b[0].split(":")[0] if b[0].startswith(".LBB") else "entry"
        per_ld = f"{valu / ld:5.1f} VALU per row load" if ld else "no row loads in the block"
        print(f"  {name:10s} row loads {ld:2d}  lookups {ds:3d}  v_perm {perm:3d}  VALU {valu:4d}  SALU {salu:3d}  "
              f"VALU per lookup {valu / ds:4.2f}  {per_ld}  vmcnt waits {','.join(waits) or '-'}")
