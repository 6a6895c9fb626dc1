#!/bin/bash
# Round-2 rocprofv3 evidence (run on the GPU box; writes under gpurun_out/prof_r02/, tools/collect_profiles_r02.py copies the
# summaries into profiles/).  Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT; mkdir -p $OUT
cd /tmp
HB="--no-cpu-baseline --headline-only --no-sync-probe --repeats 3"
# 1. headline (configs[1]): kernel durations of the timed launches only (no host-pointer calls in the trace)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h_trace -- python3 $R/bench.py $HB --steps 2000 --warmup 200 > $OUT/h_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/h_fetch -- python3 $R/bench.py $HB --steps 200 --warmup 20 > $OUT/h_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/h_write -- python3 $R/bench.py $HB --steps 200 --warmup 20 > $OUT/h_write.log 2>&1
# 2. configs[2]: B = 128, O = 50
C2="--batch 128 --obstacles 50 --steps 40 --warmup 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -- python3 $R/bench.py $HB $C2 > $OUT/c2_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -- python3 $R/bench.py $HB $C2 > $OUT/c2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c2_write -- python3 $R/bench.py $HB $C2 > $OUT/c2_write.log 2>&1
# 3. the persistent solver kernel: traffic per solve (two evaluation phases of the sample problem)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_trace -- python3 $R/tools/solve_once.py 50 > $OUT/s_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/s_fetch -- python3 $R/tools/solve_once.py 50 > $OUT/s_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/s_write -- python3 $R/tools/solve_once.py 50 > $OUT/s_write.log 2>&1
# 4. P1 chain kernel: SQ counters at B = 1 and B = 128
for B in 1 128; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1_sqa_B$B -- python3 $R/tools/p1_once.py $B > $OUT/p1_sqa_B$B.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1_sqb_B$B -- python3 $R/tools/p1_once.py $B > $OUT/p1_sqb_B$B.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1_trace_B$B -- python3 $R/tools/p1_once.py $B > $OUT/p1_trace_B$B.log 2>&1
done
cd $R
for f in $OUT/*.log; do echo "== $f"; tail -2 $f | cut -c1-300; done
find $OUT -name "*.csv" | wc -l
