#!/bin/bash
# the 200-frame stream on two builds of the backend (LIBS, relative to the repo), alternating, three rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do for lib in $LIBS; do
  O3DS_BACKEND_LIB=$R/$lib python scripts/bench_stream.py --frames 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['scans_per_sec'],1), round(d['mapping_only_scans_per_sec'],1), d.get('ms_per_scan'))"
done; done
