#!/bin/bash
# the launch that ends the loop reports to the host (trailing launches drain behind it): registration tests, configs[1], the stream twice
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4g; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_sharded_gpu.py tests/test_pipeline_gpu.py tests/test_repro_gpu.py -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 2>/dev/null | tail -1 > $OUT/m1.json; python -c "
import json; d=json.load(open('$OUT/m1.json')); print('m1', round(d['value']), d['ms_per_step'])"
for v in 1 2; do python scripts/bench_stream.py --frames 200 2>/dev/null | tail -1 > $OUT/stream$v.json; python -c "
import json; d=json.load(open('$OUT/stream$v.json')); print('stream', {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))}, d.get('ms_per_scan'), d.get('final_pose_error_vs_truth'))"; done
