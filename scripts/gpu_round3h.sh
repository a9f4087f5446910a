#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for w in 2 4 8; do
  export O3DS_SORT_WAYS=$w
  bash scripts/gpu_stream_prof.sh | grep -E "sort_" | cut -c1-200 | sed "s/^/ways=$w  /"
  python -c "
import json;d=json.load(open('$O/stream_prof.json'));print('ways=$w', {k:round(d[k],1) for k in ('scans_per_sec','mapping_only_scans_per_sec')})"
done
