#!/usr/bin/env bash
# round-3 call 1: where the headline SYRK loses its MFMA duty cycle (ablation masks + SQ stall counters), baseline timings
export PYTHONPATH=.
O=gpurun_out/c01; mkdir -p $O
timeout 300 python tools/dev/ablate_syrk_dev.py > $O/ablate.log 2>&1
sed -i 's/for mask in (0, 1, 2, 3, 7, 0)/for mask in (0, 1, 2, 4, 3, 5, 6, 7, 16, 0)/' tools/dev/ablate_syrk_dev.py
timeout 300 python tools/dev/ablate_syrk_dev.py >> $O/ablate.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P="python tools/dev/syrk_prof_dev.py 4"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/p1 -o a -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/p2 -o b -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM -d $O/p3 -o c -- $P > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/p4 -o d -- $P > $O/p4.log 2>&1
for d in p1 p2 p3 p4; do DB=$(find $O/$d -name '*results.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py pmcany $DB syrk_tn_kernel >> $O/pmc_summary.jsonl 2>&1; done
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
timeout 300 python tools/dev/bench_potrf_dev.py 8192 > $O/potrf.log 2>&1
timeout 300 python tools/dev/bench_potrf_dev.py 2048 >> $O/potrf.log 2>&1
echo done
