#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for T in 256 512 1024 2048 4096; do echo "OSN_WGRAD_T=$T"; OSN_WGRAD_T=$T ONLY=wgrad REPS=5 python tools/micro_conv.py 2>&1 | tail -1; done
for U in 4 6 8 10 14 27; do echo "OSN_UNIT_K=$U"; OSN_UNIT_K=$U ONLY=fwd REPS=5 python tools/micro_conv.py 2>&1 | tail -1; done
