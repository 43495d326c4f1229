#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( MI355KKT_TILES_MIN_N=1 timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k potrf 2>&1 | tail -3 ) > $O/r2l.log 2>&1
for n in 8192 4096 2048 1024 512 256; do MI355KKT_TILES_MIN_N=1 timeout 120 python tools/dev/bench_potrf_dev.py $n 2>&1 | tail -1 >> $O/r2l.log; done
for n in 512 256; do MI355KKT_POTRF=streams timeout 120 python tools/dev/bench_potrf_dev.py $n 2>&1 | tail -1 >> $O/r2l.log; done
MI355KKT_TILES_MIN_N=1 timeout 120 python tools/dev/prof_tiles_dev.py 2048 2>&1 | tail -8 >> $O/r2l.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# step timeline + kernel stats of the headline workload
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r02_dense -o r02 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r2l_prof_dense.log 2>&1
DB=$(find $O/prof_r02_dense -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r02_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 3 60 > $O/r02_step_timeline.txt 2>&1
# SOC scaling / assembly step (north_star: HBM GB/s on the scaling step)
for r in 4 8 64; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r02_socp$r -o socp$r -- python bench.py --workload socp --cone-dim $r --steps 4 --warmup 2 > $O/r2l_socp$r.json 2> $O/r2l_socp$r.err
  DBS=$(find $O/prof_r02_socp$r -name '*results.db' | head -1)
  python tools/rocpd_summary.py stats $DBS $O/r02_socp${r}_kernel_stats.md > /dev/null 2>&1
done
# PMC passes (their own runs, no tracing): HBM-side traffic and MFMA utilisation of the SYRK and of the tile Cholesky
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_r02_fetch -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2l_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_r02_write -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2l_pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_r02_mfma -o m -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r2l_pmc_mfma.log 2>&1
FDB=$(find $O/pmc_r02_fetch -name '*results.db' | head -1); WDB=$(find $O/pmc_r02_write -name '*results.db' | head -1)
python tools/rocpd_summary.py pmc $FDB $WDB syrk_tn_kernel $O/r02_pmc_syrk.json 8192 16384 > $O/r2l_pmc_sum.log 2>&1
python tools/rocpd_summary.py pmc $FDB $WDB potrf_tiles_kernel $O/r02_pmc_potrf_tiles.json 8192 16384 >> $O/r2l_pmc_sum.log 2>&1
echo done
