#!/bin/bash
# a variant build of the backend: scripts/build_variant.sh <suffix> [extra hipcc flags]  ->  open3d_slam_amd/lib/libo3ds_backend_<suffix>.so
R=$(cd "$(dirname "$0")/.." && pwd); S=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=8 "$@" -o $R/open3d_slam_amd/lib/libo3ds_backend_$S.so $R/open3d_slam_amd/csrc/backend.hip
