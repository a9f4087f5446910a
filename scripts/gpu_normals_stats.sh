#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
[ -f open3d_slam_amd/lib/libo3ds_backend_stats.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DO3DS_NRM_STATS -o open3d_slam_amd/lib/libo3ds_backend_stats.so open3d_slam_amd/csrc/backend.hip || exit 1
timeout 120 python scripts/normals_stats.py 2>&1 | tail -40 | tee $OUT/normals_stats.txt
