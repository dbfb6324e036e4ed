#!/bin/bash
# Readiness of the N > 1 path on a ONE-GPU box (the 8-GPU run is the driver's):
#   1. `bench.py --gpus 2` the way the driver calls N = 1 (no launcher): self-spawn under torch.distributed.run, two ranks
#      sharing device 0 over gloo (RCCL refuses two ranks on one device), sliced gradient exchange during backward;
#   2. one-rank RCCL group, exchange in 4 slices during backward / in one collective after it;
#   3. stdout of each run = exactly one JSON line (RCCL's banner goes to stderr).
# usage: bash tools/gpu_dist_check.sh TAG
tag=${1:-dist}
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # name, env..., -- args
    name=$1; shift
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" timeout 400 python bench.py "$@" > $O/$name.out 2> $O/$name.err
    rc=$?
    python - "$O/$name.out" "$name" $rc <<'PY' | tee -a $O/summary.txt
import json, sys
path, name, rc = sys.argv[1:4]
lines = open(path).read().splitlines()
js = [l for l in lines if l.startswith("{")]
if rc != "0" or len(js) != 1:
    print("%-44s rc=%s stdout lines=%d json lines=%d" % (name, rc, len(lines), len(js)))
else:
    d = json.loads(js[0])
    print("%-44s rc=0 stdout lines=%d  n_gpus=%d  %.3f ms/step  %.4g voxels/s  loss %.6f  comm=%s" % (
        name, len(lines), d["n_gpus"], d["ms_per_step"], d["value"], d["loss"], json.dumps(d.get("comm"))))
PY
}
: > $O/summary.txt
common="--steps 20 --warmup 6 --train-only --no-cpu-baseline --no-kernel-events --detail $O/detail.json"
run one_rank_no_group                          -- $common
run one_rank_rccl_4_slices                     -- --dist-single $common
run one_rank_rccl_1_collective OSN_GRAD_SEGMENTS=1 -- --dist-single $common
run two_ranks_one_device_gloo_self_spawn OSN_BENCH_ONE_DEVICE=1 OSN_DIST_BACKEND=gloo -- --gpus 2 $common
run two_ranks_one_device_gloo_1_collective OSN_BENCH_ONE_DEVICE=1 OSN_DIST_BACKEND=gloo OSN_GRAD_SEGMENTS=1 -- --gpus 2 $common
tail -5 $O/two_ranks_one_device_gloo_self_spawn.err
