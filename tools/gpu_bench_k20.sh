#!/bin/bash
# dev: the driver's command line (--steps 20 --warmup 5) and the default one, a few times each
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps 20:', round(d['value']), 'iters/s', round(d['ms_per_step']*1e3,3), 'us/step, events', round(d['roofline']['launch_us'],3), 'us, wall min/med/max', [round(x*1e3,1) for x in d['timing']['wall_ms_min_med_max']])"
done
python bench.py --no-cpu-baseline --headline-only 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps 2000:', round(d['value']), 'iters/s', round(d['ms_per_step']*1e3,3), 'us/step, events', round(d['roofline']['launch_us'],3))"
