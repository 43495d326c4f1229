"""Maps the address of a GPU memory fault to the device allocation next to it.

The HIP runtime reports a fault as `Memory access fault by GPU node-N ... on address 0x...` and abort()s; a test session run with
MI355KKT_TEST_ABORT_DUMP=<file> (tests/conftest.py -> mi355kkt_test_install_abort_dump, csrc/devmem.cpp) leaves the ring of the
library's last allocations / releases in <file>.  This prints the allocations that were live at the end and lie closest to the
address (under MI355KKT_ALLOC_GUARD the faulting page is the one right behind its owner).

    python tools/alloc_owner.py <dump file> <hex address> [how many]
"""
import sys


def main():
    path, addr = sys.argv[1], int(sys.argv[2], 16)
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    live = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        k, op, p, nbytes, site = line.split(None, 4)
        p = int(p, 16)
        if int(op) == 2:
            live.pop(p, None)
        else:
            live[p] = (int(nbytes), site.strip(), int(k), int(op))
    rows = []
    for p, (nbytes, site, k, op) in live.items():
        end = p + nbytes
        dist = 0 if p <= addr < end else (addr - end if addr >= end else p - addr)
        rows.append((dist, p, nbytes, site, k, "behind" if addr >= end else ("inside" if addr >= p else "in front")))
    rows.sort()
    print("%d live allocations; address %#x" % (len(live), addr))
    for dist, p, nbytes, site, k, where in rows[:top]:
        print("  %#x + %d bytes (event %d, %s): address is %s, %d bytes from its %s" % (
            p, nbytes, k, site, where, dist, "end" if where == "behind" else "start"))


if __name__ == "__main__":
    main()
