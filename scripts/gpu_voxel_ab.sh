#!/bin/bash
# A/B of VoxelDownSample by hashing (default) against the sort-based path (O3DS_VOXEL_SORT=1): parity tests of everything that uses it,
# then the scan stream of the bench (200 frames) both ways.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_preprocess_map_gpu.py tests/test_pipeline_gpu.py tests/test_repro_gpu.py tests/test_host_adapter.py -m gpu -q -x --timeout 400 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -30 > $OUT/voxel_ab_pytest.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/voxel_ab_pytest.log
for v in hash sort; do
  if [ $v = sort ]; then export O3DS_VOXEL_SORT=1; else unset O3DS_VOXEL_SORT; fi
  timeout 400 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-f64 --concurrent 0 2>/dev/null | tail -1 > $OUT/voxel_ab_$v.json
done
tail -3 $OUT/voxel_ab_pytest.log
python - <<'PY'
import json, os
o = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/"
for v in ("hash", "sort"):
    try:
        d = json.load(open(o + f"voxel_ab_{v}.json"))
        s = d["scans_per_sec"]
        print(v, {k: s[k] for k in s if not isinstance(s[k], (dict, list))})
        for row in s.get("calls", []):
            print("   ", row)
    except Exception as e:
        print(v, "unreadable", e)
PY
