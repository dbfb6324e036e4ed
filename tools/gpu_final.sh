#!/bin/bash
# Round-end evidence in one call: full GPU tests + smoke, the default bench line (with the CPU baseline), the same bench command
# under rocprofv3 (whole-run kernel stats + the dominant kernel's dispatch durations), per-step kernel stats of the training
# step, and the PMC traffic passes of the tile-list kernels.   usage: gpu_final.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export TMPDIR=/tmp
t0=$(date +%s)
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - t0 )) s"; tail -2 $O/pytest_gpu.log; fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cp bench_detail.json $O/bench_detail.json 2>/dev/null; fi
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/whole -o trace -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python $R/tools/rocpd_stats.py $O/whole/trace_results.db > $O/kernel_stats_whole_run.csv
python $R/tools/rocpd_stats.py $O/whole/trace_results.db --dispatches spconv_tl_kernel 400 > $O/tl_dispatches.txt
rm -rf $O/whole
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
# the dominant launch SHAPE by itself: a step issues its 13 spconv_tl_kernel<3, 3> launches in a fixed order (tools/tl_launch_sequence.py:
# positions 3-5 = forward, 6-8 = input gradient of the 3^3 96 -> 96 convs on the 100 999-row map), so folding the dispatches modulo 13
# separates the shapes without any help from the application
python $R/tools/tl_launch_sequence.py MinkUNet18A 768 3 3 > $O/tl33_labels.json
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --by-position spconv_tl_kernelILi3ELi3E 13 $O/tl33_labels.json > $O/tl33_by_position.txt
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --by-position wgrad_tl_kernelILi3ELi3E 7 > $O/wgrad33_by_position.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
# where the main stream idles in a step and what the other streams run meanwhile (host timing under the profiler is distorted: read the
# GPU-side waits -- joins, tails -- not the host-bound gaps)
python $R/tools/gap_census.py $O/prof/trace_results.db 3 40 > $O/gap_census.txt 2>&1
rm -rf $O/prof
cat $O/groups.txt | head -24
cd $R
bash tools/pmc_tl.sh > $O/pmc_tl.log 2>&1; cp gpurun_out/pmc_tl/summary.txt $O/pmc_tl_summary.txt 2>/dev/null; rm -rf gpurun_out/pmc_tl/*_p*; tail -3 $O/pmc_tl.log
bash tools/pmc_tl_sq.sh > $O/pmc_tl_sq.log 2>&1; cp gpurun_out/pmc_tl_sq/summary.txt $O/pmc_tl_sq_summary.txt 2>/dev/null; rm -rf gpurun_out/pmc_tl_sq/*_p*; tail -3 $O/pmc_tl_sq.log
cat $O/tl33_by_position.txt
# what each call-site edit of INTEGRATION.md section 6 buys: the headline step with one of them undone at a time
cd $R
for fl in "" "--torch-loss" "--torch-adam" "--no-prefetch-maps" "--no-prefetch-maps --no-prefetch-pyramid" "--torch-loss --torch-adam --no-prefetch-maps --no-prefetch-pyramid"; do
  timeout 200 python bench.py --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/ladder_detail.json $fl > $O/ladder.tmp 2>/dev/null
  python -c "
import json
d=[json.loads(l) for l in open('$O/ladder.tmp') if l.startswith('{')][-1]; print('%-80s %.3f ms/step' % ('[$fl]', d['ms_per_step']))" >> $O/call_site_ladder.txt
done
timeout 200 python bench.py --dist-single --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/ladder_detail.json > $O/ladder.tmp 2>$O/dist_single.err
python -c "
import json
d=[json.loads(l) for l in open('$O/ladder.tmp') if l.startswith('{')][-1]; print('%-80s %.3f ms/step  comm %s' % ('[--dist-single: one-rank RCCL group, sliced exchange]', d['ms_per_step'], d.get('comm')))" >> $O/call_site_ladder.txt
OSN_GRAD_SEGMENTS=1 timeout 200 python bench.py --dist-single --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/ladder_detail.json > $O/ladder.tmp 2>>$O/dist_single.err
python -c "
import json
d=[json.loads(l) for l in open('$O/ladder.tmp') if l.startswith('{')][-1]; print('%-80s %.3f ms/step' % ('[--dist-single OSN_GRAD_SEGMENTS=1: one collective after backward]', d['ms_per_step']))" >> $O/call_site_ladder.txt
cat $O/call_site_ladder.txt
[ -f $O/bench.json ] && python -c "
import json
for l in open('$O/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline'])"
