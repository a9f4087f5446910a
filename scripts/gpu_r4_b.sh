#!/bin/bash
# Round 4, GPU visit B: ICP + croppers + pipeline tests, configs[1] alone, one frame of the stream as the GPU saw it
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4b
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_reference_golden_gpu.py tests/test_edge_parity_gpu.py tests/test_pipeline_gpu.py tests/test_preprocess_map_gpu.py -m gpu -q 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -40 > $OUT/pytest.log
tail -3 $OUT/pytest.log
M1="python bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 --large-map 0 --no-host-seam"
for fl in 0.5 0 1.0; do
O3DS_FIRST_LOOK=$fl $M1 --steps 100 --warmup 10 2>/dev/null | grep '^{' | tail -1 > $OUT/m1_fl$fl.json
python -c "
import json; d=json.load(open('$OUT/m1_fl$fl.json')); print('M1 first_look=$fl', round(d['value']), 'it/s', round(d['ms_per_step']*1e3,1), 'us/step frac', round(d['roofline']['frac'],4))"
done
for cfg in "1 0.5" "0 0.5" "1 0" "1 1.0" "1 0.5"; do set -- $cfg; O3DS_P0_SETS=$1 O3DS_FIRST_LOOK=$2 python scripts/bench_stream.py --frames 100 2>/dev/null | tail -1 > $OUT/stream_$1_$2.json; python -c "
import json; d=json.load(open('$OUT/stream_$1_$2.json')); print('stream p0_sets=$1 first_look=$2', {k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))})"; done
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_stream
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stream -o s -- python $R/scripts/bench_stream.py --frames 100 > /dev/null 2>&1
python $R/scripts/prof_summary.py $OUT/prof_stream/s_results.db $OUT/rocprof_stats_stream.txt > /dev/null
python $R/scripts/prof_sequence.py $OUT/prof_stream/s_results.db $OUT/stream_frame_sequence.txt 60 | grep "icp_fused\|dispatches" 
rm -rf $OUT/prof_stream
