#!/bin/bash
# the 200-frame stream on two builds, alternating: point-to-plane (bench_stream.py) and the shipped configuration (GeneralizedIcp, ratio 0.3)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do for v in $LIBS; do
  lib=$R/open3d_slam_amd/lib/libo3ds_backend_$v.so; [ "$v" = default ] && lib=$R/open3d_slam_amd/lib/libo3ds_backend.so
  echo "$v: $(O3DS_BACKEND_LIB=$lib python scripts/bench_stream.py --frames ${FRAMES:-200} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['scans_per_sec'],1))") scans/s | shipped: $(O3DS_BACKEND_LIB=$lib python - <<PY 2>/dev/null
import sys; sys.path.insert(0, "$R")
import bench
from open3d_slam_amd import backend
scans = bench.make_stream(${FRAMES:-200})
be = backend.Backend(0); bench.run_stream(be, scans[:12], shipped=True); be.close()
be = backend.Backend(0); r = bench.run_stream(be, scans, shipped=True); be.close()
print(round(r["scans_per_sec"], 1))
PY
) scans/s"
done; done
