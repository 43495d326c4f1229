#!/usr/bin/env python3
"""Static resource usage of every kernel in the gfx950 code objects (no GPU needed): VGPR / AGPR / SGPR counts, LDS
(static), scratch bytes and spill counts, read from the code-object metadata.

    python tools/kernel_resources.py [out.txt]

Compiles each csrc/*.hip for the device only (hipcc --cuda-device-only), unbundles the gfx950 ELF and parses the
amdhsa.kernels notes.  Used to check that the hot kernels keep their intended occupancy (syrk_tn_kernel: 256 registers
= 2 waves per SIMD; its 6 spilled VGPRs are stored once before and reloaded once after the K loop) and that nothing
else spills to scratch."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
PAT = re.compile(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                 r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                 r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", re.S)


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    tmp = tempfile.mkdtemp()
    print("%-12s %-58s %5s %5s %5s %7s %7s %6s %6s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "sspill",
                                                      "vspill"), file=out)
    for src in sorted(glob.glob(os.path.join(ROOT, "cvxopt_amd", "csrc", "*.hip"))):
        base = os.path.basename(src)[:-4]
        co, elf = os.path.join(tmp, base + ".co"), os.path.join(tmp, base + ".elf")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", src,
                               "-o", co], stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
        for m in PAT.finditer(notes):
            ag, lds, name, priv, sg, sgs, vg, vgs = m.groups()
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem).replace("mi355kkt::", "").replace("void ", "")[:58]
            print("%-12s %-58s %5s %5s %5s %7s %7s %6s %6s" % (base, dem, vg, ag, sg, lds, priv, sgs, vgs), file=out)


if __name__ == "__main__":
    main()
