#!/bin/bash
# the one-ulp wobble of two-handle runs under A/B switches: how many of 6 two-handle runs per process differ from the one-handle run
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 DEBUG_AB=1
run() { echo "== $*"; for i in 1 2 3; do env "$@" timeout 200 python scripts/debug_wobble.py 200 6 2>&1 | grep -c "^two handles.*result: (" ; done | tr '\n' ' '; echo; }
run X=1
if [ -n "$WOBBLE_ALL" ]; then
run O3DS_NO_PERSISTENT_MAP=1
run O3DS_ICP_SETS=0
run O3DS_ICP_MODE=launch
run O3DS_SYNC_MASK=65535
run O3DS_ALWAYS_CLEAR=1
else
run X=2
fi
