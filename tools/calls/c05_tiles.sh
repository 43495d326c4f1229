#!/usr/bin/env bash
# round-3 call 5: where the streamed tile Cholesky spends its time at n = 8192 (all tile stamps, MFMA duty, stall counters)
export PYTHONPATH=.
O=gpurun_out/c05; mkdir -p $O
timeout 300 python tools/dev/dump_tiles_dev.py 8192 $O/tiles_8192.npy > $O/dump.log 2>&1
timeout 300 python tools/dev/dump_tiles_dev.py 4096 $O/tiles_4096.npy >> $O/dump.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P="python tools/dev/bench_potrf_dev.py 8192"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/p1 -o a -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 -o b -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/p3 -o c -- $P > $O/p3.log 2>&1
for d in p1 p2 p3; do DB=$(find $O/$d -name '*results.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py pmcany $DB potrf_tiles_kernel >> $O/pmc_summary.jsonl 2>&1; done
rm -rf $O/p1 $O/p2 $O/p3
echo done
