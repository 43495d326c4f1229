"""tools/alloc_owner.py on a synthetic allocation ring (the format csrc/devmem.cpp's SIGABRT handler writes): the owner of a fault
address is the live block it lies in or right behind; released blocks are not candidates."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alloc_owner_names_the_live_block_next_to_the_address(tmp_path):
    dump = tmp_path / "ring.txt"
    dump.write_text("\n".join([
        "# mi355kkt allocation events 0..6 (op 1 alloc, 2 free, 3 guarded alloc)",
        "0 1 0x7f0000000000 4096 /root/repo/cvxopt_amd/csrc/capi.hip:536",
        "1 3 0x7f00001ff000 4096 /root/repo/cvxopt_amd/csrc/capi.hip:541",
        "2 1 0x7f0000400000 1048576 /root/repo/cvxopt_amd/csrc/potrf.hip:1249",
        "3 2 0x7f0000000000 0 ?:0",
        "4 3 0x7f00005ffe00 512 /root/repo/cvxopt_amd/csrc/sparse_chol.hip:1433",
        "5 2 0x7f0000400000 0 ?:0",
        ""]))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "alloc_owner.py"), str(dump), "0x7f0000600000", "3"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith("2 live allocations")                 # events 0 and 2 were released
    # the fault page starts exactly where the guarded 512-byte block of event 4 ends
    assert "event 4" in lines[1] and "sparse_chol.hip:1433" in lines[1] and "behind" in lines[1] and " 0 bytes from its end" in lines[1]
    assert "event 1" in lines[2]
