#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
export TMPDIR=/tmp
ab() {
  tag=$1; shift
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only "$@" > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
for rep in 1 2; do
ab pyr$rep
ab nopf$rep --no-prefetch-pyramid
ab full$rep --prefetch-maps
done
tail -n 5 $O/ab.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
rm -rf $O/prof
tail -n 22 $O/streams.txt
