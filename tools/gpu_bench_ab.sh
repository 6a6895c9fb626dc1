#!/bin/bash
# dev: A/B of the whole bench line (headline, configs[2], configs[4]) between libraries / settings, interleaved rounds on ONE box.
#   tools/gpu_bench_ab.sh <rounds> <variant>...     variant = <lib name or 'tree'>[@VAR=value[@VAR=value...]]
#   (lib name: armour_amd/lib/libarmour_hip_<name>.so; VAR=value is exported for that variant's runs only)
rounds=$1; shift
for i in $(seq $rounds); do for v in "$@"; do
  (
  IFS=@ read -r -a parts <<< "$v"
  lib=${parts[0]}
  for kv in "${parts[@]:1}"; do export "$kv"; done
  if [ $lib != tree ]; then export ARMOUR_HIP_LIB=$PWD/armour_amd/lib/libarmour_hip_$lib.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-sync-probe --repeats 5 $BENCH_AB_FLAGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); oc=d.get('other_configs',{})
print('$v', 'headline us', round(d['roofline']['launch_us'],3), 'step us', round(d['ms_per_step']*1e3,3), ' | '.join(f\"{k.split(':')[0]} {v['launch_us']:.2f} us, P1 {v['p1_set_problems_ms_per_problem']:.4f} ms/problem\" for k,v in oc.items()))"
  )
done; done
