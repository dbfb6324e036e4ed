#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
for bm in 32 48 64 80; do
  echo "== l1 BM=$bm"; BM=$bm SHAPES=l1 REPS=20 timeout 200 python tools/micro_tl.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d.get('bm'), 'tl %.1f us  %.1f TF  x6 %.1f' % (d['tl_us'], d['tl_TF'], d['x6_us']))"
done
for bm in 64 72 80 88; do
  echo "== hot BM=$bm"; BM=$bm SHAPES=hot REPS=20 timeout 200 python tools/micro_tl.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d.get('bm'), 'tl %.1f us  %.1f TF' % (d['tl_us'], d['tl_TF']))"
done
