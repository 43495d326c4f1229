#!/usr/bin/env python3
"""Static check of the fence-free streaming hand-off of the tile Cholesky (csrc/potrf.hip, DESIGN 4a').

A tile on the chain announces column block cb - 1 of itself one block late with

    st_wt x 4 (block cb)  ->  s_waitcnt vmcnt(4)  ->  s_barrier  ->  relaxed agent-scope store of the progress word

i.e. "everything but this block's four write-through stores has reached memory".  The literal 4 is right only if (a) exactly
four vector-memory instructions were issued since block cb - 1's stores, (b) they are the sc1 (write-through) stores themselves
-- a scratch spill or a load slipped in by the compiler would be counted by vmcnt too -- and (c) a wave's stores are acknowledged
in issue order (gfx9 family: vmcnt decrements in order for stores; potrf.hip refuses to compile for anything but gfx942 / gfx950).
(a) and (b) are properties of the COMPILED code, so they are checked on it: this script compiles potrf.hip to gfx950 assembly and
requires that the four vector-memory instructions in front of every `s_waitcnt vmcnt(4)` are `global_store_dwordx2 ... sc1` AND that
they sit in the wait's own basic block (no label, no branch in between: the source keeps the announce path branch-free for this).

    python tools/check_stream_isa.py            # exit status 0 = every site verified
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r"^\s+(global_|flat_|buffer_|scratch_|tbuffer_|image_)\w+")


LABEL = re.compile(r"^(\.LBB\S+|_Z\S*):")
BRANCH = re.compile(r"^\s+s_(c?branch|setpc|swappc|endpgm|call)")


def check(asm_path):
    lines = open(asm_path).read().split("\n")
    sites, bad = 0, []
    for i, l in enumerate(lines):
        # only the hand-written waits (inline asm: bracketed by ;;#ASMSTART / ;;#ASMEND); the compiler's own vmcnt(4) waits for
        # loads are its business
        if "s_waitcnt vmcnt(4)" not in l or i == 0 or "#ASMSTART" not in lines[i - 1]:
            continue
        sites += 1
        found, why = [], None
        j = i - 1
        # backwards INSIDE the basic block of the wait: a label (another path joins here) or a branch (this code may be skipped)
        # before four vector-memory instructions have been seen means the count is not a property of straight-line code
        while j >= 0 and len(found) < 4:
            if VMEM.match(lines[j]):
                found.append(lines[j].strip())
            elif LABEL.match(lines[j]) or BRANCH.match(lines[j]):
                why = "control flow (%s) between the wait and its four stores" % lines[j].strip()
                break
            j -= 1
        ok = why is None and len(found) == 4 and all(f.startswith("global_store_dwordx2") and f.rstrip().endswith("sc1") for f in found)
        if not ok:
            bad.append((i + 1, why or found))
    return sites, bad


def digest(paths, hipcc):
    h = hashlib.sha256()
    for p_ in paths:
        h.update(open(p_, "rb").read())
    try:
        h.update(subprocess.check_output([hipcc, "--version"], stderr=subprocess.STDOUT))
    except (OSError, subprocess.CalledProcessError):
        pass
    return h.hexdigest()


def main():
    csrc = os.path.join(ROOT, "cvxopt_amd", "csrc")
    src = os.path.join(csrc, "potrf.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the verdict is a function of the source, its headers and the compiler: remembered next to the objects, so that a build()
    # that compiled nothing does not compile potrf.hip again just to check it
    stamp = os.path.join(csrc, ".obj", "check_stream_isa.ok")
    key = digest([src, os.path.join(csrc, "kkt_common.h"), os.path.join(csrc, "knobs.h"), os.path.abspath(__file__)], hipcc)
    if "--force" not in sys.argv and os.path.exists(stamp) and open(stamp).read().split("\n")[0] == key:
        print("check_stream_isa: %s (unchanged sources and compiler: verdict of the last check)" % open(stamp).read().split("\n")[1])
        return 0
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "potrf.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            print(r.stdout)
            print("check_stream_isa: %s failed to compile potrf.hip to assembly" % hipcc)
            return 1
        sites, bad = check(out)
    if sites == 0:
        print("check_stream_isa: no `s_waitcnt vmcnt(4)` in the compiled potrf.hip -- the streaming announce is gone?")
        return 1
    for line, found in bad:
        print("check_stream_isa: line %d: the four vector-memory instructions before vmcnt(4) are not this block's sc1 stores: %s"
              % (line, found))
    msg = "%d announce sites, %d verified" % (sites, sites - len(bad))
    print("check_stream_isa: " + msg)
    if not bad and os.path.isdir(os.path.dirname(stamp)):
        with open(stamp, "w") as f:
            f.write(key + "\n" + msg + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
