#!/usr/bin/env bash
# round-5 call 14: the reference-generated fixture of the irregular config-4 stand-in through the device loop
export PYTHONPATH=.
O=gpurun_out/r5c14; mkdir -p $O
( MI355KKT_PARITY_REPORT=$PWD/$O/parity.json timeout 150 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k elasticity ) > $O/log.txt 2>&1
echo "rc=$? $(tail -1 $O/log.txt | cut -c1-200)" > $O/summary.txt
grep -h "assert\|Error" $O/log.txt | head -5 >> $O/summary.txt
cat $O/parity.json >> $O/summary.txt 2>/dev/null
cat $O/summary.txt
