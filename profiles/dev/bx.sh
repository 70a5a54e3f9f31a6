#!/bin/bash
# quick extraction bench summary: bash profiles/dev/bx.sh [tag]
python bench.py --no-window --no-cpu-baseline > gpurun_out/bx_$1.json 2>gpurun_out/bx_$1.err; python -c "
import json; d=json.load(open('gpurun_out/bx_$1.json'))
print('C2', d['ms_per_step'], d['stages_ms'], 'soa', d['soa_input']['ms_per_step'], 'frac', d['roofline']['frac'])
print('firing', d['firing_order']['ms_per_step'], d['firing_order']['stages_ms'])
c=d['cloud_10m']; print('10m', c['ms_per_step'], c['stages_ms'], 'soa', c['soa_input']['ms_per_step'], c['soa_input']['stages_ms'])
b=d['batched_10x_c2']; print('batch', b['aos']['ms_per_sweep'], b['soa']['ms_per_sweep'], 'und+ext', b['undistort_then_extract']['ms_per_sweep'])
"
