#!/bin/bash
# rocprofv3 evidence of one round (run ON THE GPU BOX):   bash tools/profile.sh <tag>      e.g. tools/profile.sh r05
# Writes under gpurun_out/prof_<tag>/; `python tools/collect_profiles.py <tag>` (here, afterwards) copies the summaries into profiles/<tag>_*.
# Counter passes are separate runs with --pmc only (no trace domains), one counter group per pass, each under its own timeout, the program
# itself after `--` -- as the pool requires.
set -u
TAG=${1:?usage: tools/profile.sh <tag>}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
HB="--no-cpu-baseline --headline-only --no-sync-probe --repeats 3"
run() { d=$1; shift; timeout -k 10 240 "$@" > $OUT/$d.log 2>&1; echo "$d rc=$?"; }
three() {   # kernel stats + FETCH_SIZE + WRITE_SIZE/L2 passes of one command:  three <name> <program and arguments>
  n=$1; shift
  run ${n}_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${n}_trace -- "$@"
  run ${n}_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${n}_fetch -- "$@"
  run ${n}_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${n}_write -- "$@"
}
# 1. headline (configs[1]): kernel durations of the timed launches only (no host-pointer calls in the trace)
run h_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h_trace -- python3 $R/bench.py $HB --steps 2000 --warmup 200
run h_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/h_fetch -- python3 $R/bench.py $HB --steps 200 --warmup 20
run h_write rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/h_write -- python3 $R/bench.py $HB --steps 200 --warmup 20
# 2. configs[2]: B = 128, O = 50;  2b. configs[4]: Fetch, O = 100;  2c. its 8-factor form (128-bit keys: the ABI is per process)
three c2 python3 $R/bench.py $HB --batch 128 --obstacles 50 --steps 40 --warmup 4
three c4 python3 $R/tools/workload.py c4 40
export ARMOUR_KEY128=1
three c48 python3 $R/tools/workload.py c4_8f 40
unset ARMOUR_KEY128
# 3. the persistent solver kernel (two evaluation phases of the sample problem per launch)
three s python3 $R/tools/workload.py solve 50
# 4. the reach-set build at B = 1 (per-step kernel) and B = 128 (time-vectorised kernel): kernel stats, traffic, L2 hit rate, SQ counters
for B in 1 128; do
  run p1_trace_B$B rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p1_trace_B$B -- python3 $R/tools/workload.py p1 $B
  run p1_fetch_B$B rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1_fetch_B$B -- python3 $R/tools/workload.py p1 $B
  run p1_write_B$B rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p1_write_B$B -- python3 $R/tools/workload.py p1 $B
  run p1_l2_B$B rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p1_l2_B$B -- python3 $R/tools/workload.py p1 $B
  run p1_sqa_B$B rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p1_sqa_B$B -- python3 $R/tools/workload.py p1 $B
  run p1_sqb_B$B rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1_sqb_B$B -- python3 $R/tools/workload.py p1 $B
done
# 5. a batch of 128 worlds at O = 50: the half-space kernels of the build;  6. the culled row test at configs[2]
run pl_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pl_fetch -- python3 $R/bench.py $HB --batch 128 --obstacles 50 --steps 40 --warmup 4
run cull_trace rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cull_trace -- python3 $R/tools/workload.py cull 128 50
cd $R
find $OUT -name "*.csv" | wc -l
find $OUT -name "*kernel_trace.csv" -size +2M -delete   # (the per-dispatch traces of the long runs are not needed, the stats and counter tables are)
du -sh $OUT
