#!/bin/bash
# usage: tools/ab_wall.sh ROUNDS "ENV=VAL ..." "ENV=VAL ..." ...   -> wall-clock ms/step of 40 training steps per
# configuration, configurations interleaved ROUNDS times (box-to-box and minute-to-minute drift is +-5 %)
rounds=$1; shift
out=/root/repo/gpurun_out/abw
mkdir -p $out
for r in $(seq 1 $rounds); do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout 300 python /root/repo/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $out/w_${i}_$r.json 2> $out/w_${i}_$r.err
    echo "round $r cfg $i [$cfg]: $(python -c "import json;print(round(json.loads(open('$out/w_${i}_$r.json').read().strip().splitlines()[-1])['ms_per_step'],3))")"
  done
done
