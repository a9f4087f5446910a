#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in NO_WARM QHI_LATE OLD_TABLES; do
  echo "== $v"; O3DS_BACKEND_LIB=$R/open3d_slam_amd/lib/libdbg_$v.so O3DS_ICP_SETS=0 python scripts/debug_sets.py 1 2>&1 | grep "^prec" | cut -c1-160 | head -3
done
