"""Developer probe: per-block phase stamps of the trsv_z kernel (MI355KKT_TRSV=z) at n = 8192."""
import ctypes as C, os, sys
os.environ["MI355KKT_TRSV"] = "z"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth
L = _capi.lib()
n, m = 8192, 16384
pr = synth.dense_qp(n, m, seed=0)
W = synth.random_scaling(pr['dims'], seed=1, spread=1.0)
f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
solve = f(W, pr['P'])
rng = np.random.default_rng(0)
nb = n // 128
ts = _capi.DeviceBuffer.from_array(np.zeros(nb * 8))
for rep in range(3):
    x, z = rng.standard_normal(n), rng.standard_normal(m)
    if rep == 2:
        _capi.check(L.mi355kkt_debug_trsvz_ts(C.c_void_p(ts.ptr)), "ts")
    solve(x, np.zeros(0), z)
t = ts.to_array((nb, 8), dtype="int64", order="C")      # the LAST launch (backward solve) wrote these
_capi.check(L.mi355kkt_debug_trsvz_ts(None), "off")
np.save("gpurun_out/trsvz_stamps.npy", t)
clk = 2400.0
print("two workgroups per block row.  finisher: 0 strips issued, 3 phase C done, 4 c' arrived, 5 x_{k-1} arrived, 6 dot done, 7 published;"
      " solver: 1 phase A done, 2 c' published.  us; backward solve: block 63 first")
print("block | solver B (2-1) | finisher: C done->c' (4-3) | c'->x_{k-1} (5-4) | D (6-5) | publish (7-6) | hop: published - previous block's published")
prev = None
for k in list(range(nb - 1, nb - 10, -1)) + list(range(34, 28, -1)) + list(range(5, -1, -1)):
    d = t[k]
    hop = (d[7] - t[k + 1][7]) / clk if k + 1 < nb else 0.0
    print("%5d | %7.2f | %7.2f | %7.2f | %6.2f | %6.2f | %7.2f" % (k, (d[2] - d[1]) / clk, (d[4] - d[3]) / clk, (d[5] - d[4]) / clk,
                                                                (d[6] - d[5]) / clk, (d[7] - d[6]) / clk, hop))
print("first -> last published: %.1f us" % ((t[0][7] - t[nb - 1][7]) / clk))
print("timings", f.engine.timings())
