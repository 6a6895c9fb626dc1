#!/bin/bash
# Round-4 rocprofv3 evidence (run on the GPU box; writes under gpurun_out/prof_r04/, tools/collect_profiles_r04.py copies the
# summaries into profiles/).  Counter passes are separate runs with --pmc only (no trace domains), one counter group per pass,
# each under its own timeout, the program itself after `--` -- as the pool requires.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp
HB="--no-cpu-baseline --headline-only --no-sync-probe --repeats 3"
run() { d=$1; shift; timeout -k 10 240 "$@" > $OUT/$d.log 2>&1; echo "$d rc=$?"; }
# 1. headline (configs[1]): kernel durations of the timed launches only (no host-pointer calls in the trace)
run h_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h_trace -- python3 $R/bench.py $HB --steps 2000 --warmup 200
run h_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/h_fetch -- python3 $R/bench.py $HB --steps 200 --warmup 20
run h_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/h_write -- python3 $R/bench.py $HB --steps 200 --warmup 20
# 2. configs[2]: B = 128, O = 50
C2="--batch 128 --obstacles 50 --steps 40 --warmup 4"
run c2_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_trace -- python3 $R/bench.py $HB $C2
run c2_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c2_fetch -- python3 $R/bench.py $HB $C2
run c2_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c2_write -- python3 $R/bench.py $HB $C2
# 2b. configs[4]: Fetch, O = 100, one problem
run c4_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_trace -- python3 $R/tools/c4_once.py 40
run c4_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4_fetch -- python3 $R/tools/c4_once.py 40
run c4_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c4_write -- python3 $R/tools/c4_once.py 40
# 3. the persistent solver kernel: traffic per solve (two evaluation phases of the sample problem)
run s_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s_trace -- python3 $R/tools/solve_once.py 50
run s_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/s_fetch -- python3 $R/tools/solve_once.py 50
run s_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/s_write -- python3 $R/tools/solve_once.py 50
# 4. the reach-set build at B = 1 (per-step kernel) and B = 128 (time-vectorised kernel): kernel stats, HBM traffic, L2 hit rate, SQ counters
for B in 1 128; do
  run p1_trace_B$B rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1_trace_B$B -- python3 $R/tools/p1_once.py $B
  run p1_fetch_B$B rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1_fetch_B$B -- python3 $R/tools/p1_once.py $B
  run p1_write_B$B rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p1_write_B$B -- python3 $R/tools/p1_once.py $B
  run p1_l2_B$B rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p1_l2_B$B -- python3 $R/tools/p1_once.py $B
  run p1_sqa_B$B rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1_sqa_B$B -- python3 $R/tools/p1_once.py $B
  run p1_sqb_B$B rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1_sqb_B$B -- python3 $R/tools/p1_once.py $B
done
# 5. a batch of 128 worlds at O = 50: the half-space kernels of the build (planes, class pre-pass, sampled rows)
run pl_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pl_fetch -- python3 $R/bench.py $HB $C2
cd $R
find $OUT -name "*.csv" | wc -l
# keep what travels back small: the per-dispatch kernel traces of the long runs are not needed, the stats and counter tables are
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
