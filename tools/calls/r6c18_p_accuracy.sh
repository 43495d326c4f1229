#!/usr/bin/env bash
# round-6 call 18: solves with equality constraints against the oracle at the tightened 3 x bound (round-5 shapes + (4096, 16), (8192, 16))
export PYTHONPATH=.
O=gpurun_out/r6c18; mkdir -p $O
export MI355KKT_PARITY_REPORT=$PWD/$O/parity_report.json
timeout 1800 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -q -k "oracle or wide" > $O/pytest.txt 2>&1
tail -30 $O/pytest.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6c18/parity_report.json"))
for k,v in sorted(d.items()):
    if 'round5_solves' in k or 'round6_wide' in k:
        print(k, {a: ("%.2e" % b if isinstance(b, float) else b) for a, b in v.items()})
PY
