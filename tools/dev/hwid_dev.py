"""Developer probe: where do workgroups land?  HW_ID fields (gfx9: CU_ID [11:8], SH_ID [12], SE_ID [15:13]) and XCC_ID."""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = 4096
out = np.zeros(2 * n, dtype=np.uint32)
_capi.check(L.mi355kkt_debug_hwid(out.ctypes.data, n), "hwid")
hw, xcc = out[0::2], out[1::2] & 0xF
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
print("distinct xcc", sorted(set(xcc.tolist())), "se", sorted(set(se.tolist())), "sh", sorted(set(sh.tolist())), "cu", sorted(set(cu.tolist())))
keys = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("distinct (xcc,se,sh,cu):", len(keys))
per = collections.Counter((x, s) for (x, s, h, c) in keys)
print("CUs per (xcc,se):", sorted(per.items())[:12])
print("cu_id==0 count:", sum(1 for k in keys if k[3] == 0), " first 24 blocks (xcc,se,sh,cu):", [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(24)])
