#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4c; mkdir -p $OUT; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_repro_gpu.py -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -30 > $OUT/pytest.log; tail -3 $OUT/pytest.log
for v in 1 0 1 0; do O3DS_SHARE_PREPROCESS=$v python scripts/bench_stream.py --frames 200 2>/dev/null | tail -1 > $OUT/stream_share$v.json; python -c "
import json; d=json.load(open('$OUT/stream_share$v.json')); print('stream share=$v', {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))}, d.get('ms_per_scan'), d.get('final_pose_error_vs_truth'))"; done
