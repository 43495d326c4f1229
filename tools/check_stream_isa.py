#!/usr/bin/env python3
"""Static check of the fence-free streaming hand-off of the tile Cholesky (csrc/potrf.hip, DESIGN 4a').

A tile on the chain announces column block cb - 1 of itself one block late with

    st_wt x 4 (block cb)  ->  s_waitcnt vmcnt(4)  ->  s_barrier  ->  relaxed agent-scope store of the progress word

i.e. "everything but this block's four write-through stores has reached memory".  The literal 4 is right only if (a) exactly
four vector-memory instructions were issued since block cb - 1's stores, (b) they are the sc1 (write-through) stores themselves
-- a scratch spill or a load slipped in by the compiler would be counted by vmcnt too -- and (c) a wave's stores are acknowledged
in issue order (gfx9 family: vmcnt decrements in order for stores; potrf.hip refuses to compile for anything but gfx942 / gfx950).
(a) and (b) are properties of the COMPILED code, so they are checked on it: this script compiles potrf.hip to gfx950 assembly and
requires that the four vector-memory instructions in front of every `s_waitcnt vmcnt(4)` are `global_store_dwordx2 ... sc1`.

    python tools/check_stream_isa.py            # exit status 0 = every site verified
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r"^\s+(global_|flat_|buffer_|scratch_|tbuffer_|image_)\w+")


def check(asm_path):
    lines = open(asm_path).read().split("\n")
    sites, bad = 0, []
    for i, l in enumerate(lines):
        # only the hand-written waits (inline asm: bracketed by ;;#ASMSTART / ;;#ASMEND); the compiler's own vmcnt(4) waits for
        # loads are its business
        if "s_waitcnt vmcnt(4)" not in l or i == 0 or "#ASMSTART" not in lines[i - 1]:
            continue
        sites += 1
        found = []
        j = i - 1
        while j >= 0 and len(found) < 4:
            if VMEM.match(lines[j]):
                found.append(lines[j].strip())
            if re.match(r"^_Z\S*:", lines[j]):          # start of the function: fewer than four
                break
            j -= 1
        ok = len(found) == 4 and all(f.startswith("global_store_dwordx2") and f.rstrip().endswith("sc1") for f in found)
        if not ok:
            bad.append((i + 1, found))
    return sites, bad


def main():
    src = os.path.join(ROOT, "cvxopt_amd", "csrc", "potrf.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "potrf.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src,
                               "-o", out], stderr=subprocess.DEVNULL)
        sites, bad = check(out)
    if sites == 0:
        print("check_stream_isa: no `s_waitcnt vmcnt(4)` in the compiled potrf.hip -- the streaming announce is gone?")
        return 1
    for line, found in bad:
        print("check_stream_isa: line %d: the four vector-memory instructions before vmcnt(4) are not this block's sc1 stores: %s"
              % (line, found))
    print("check_stream_isa: %d announce sites, %d verified" % (sites, sites - len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
