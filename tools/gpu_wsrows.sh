#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
mkdir -p gpurun_out/r4p
for v in 4096 8192 16384; do
OSN_WS_MAX_ROWS=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > gpurun_out/r4p/b_$v.json 2>/dev/null
python -c "
import json
for l in open('gpurun_out/r4p/b_$v.json'):
    if l.startswith('{'):
        d=json.loads(l); p=d['phases']; print('WS_MAX_ROWS=$v', 'step', round(d['ms_per_step'],3), 'batch8_step', round(p['batch8_step']['ms'],2), 'batch8_inf', round(p['batch8_inference']['ms'],2), 'l235k_step', round(p['l235k_34c_step']['ms'],2), 'l235k_inf', round(p['l235k_34c_inference']['ms'],2), 'inf', round(p['inference_fwd']['ms'],3))"
done
