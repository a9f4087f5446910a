#!/bin/bash
# One GPU-box visit that refreshes every piece of evidence kept under profiles/: parity tests, smoke, the bench line (+ rocprofv3 kernel
# stats of the same command), the other bench configurations, normals parity / per-wavefront statistics.  Outputs under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{ echo "nproc=$(nproc)"; lscpu | grep -E "Model name|Socket|Core|Thread|MHz" ; } > $OUT/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rA --timeout 400 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -150 > $OUT/pytest_gpu.log
echo "pytest rc=${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
timeout 300 python bench.py --config 3u --no-cpu-baseline --no-f64 2>/dev/null | tail -1 > $OUT/bench_config3u_n1.json
timeout 300 python bench.py --config 4 2>/dev/null | tail -1 > $OUT/bench_config4_n1.json
O3DS_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config 3u --steps 20 --warmup 2 --no-f64 2>/dev/null | tail -1 > $OUT/bench_config3u_2ranks_one_gpu_gloo.json
O3DS_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --config 4 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_config4_2ranks_one_gpu_gloo.json
timeout 300 python scripts/check_normals.py > $OUT/check_normals.log 2>&1
( cd open3d_slam_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DO3DS_NRM_CHECK -DO3DS_NRM_PHASES -o lib/libo3ds_check.so csrc/backend.hip 2>/dev/null )
O3DS_NRM_PHASES=1 O3DS_BACKEND_LIB=$R/open3d_slam_amd/lib/libo3ds_check.so timeout 200 python scripts/normals_stats.py 3.0 20 > $OUT/normals_stats.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --no-cpu-baseline > $OUT/rocprof_bench.json 2> $OUT/rocprof.err
echo "rocprof rc=$?" >> $OUT/rocprof.err
python $R/scripts/prof_summary.py $OUT/prof/bench_results.db $OUT/rocprof_stats.txt > /dev/null
bash $R/scripts/gpu_stream_prof.sh > /dev/null 2>&1
# configs[1] alone (the registrations `value` and `roofline` are quoted on): the fused kernel's average here is the bench line's avg_launch_us
# once the one prologue-only launch per registration (the ~6 us dispatches of the detail list) is set aside
cd /tmp; rm -rf $OUT/prof_m1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_m1 -o m1 -- python $R/bench.py --no-cpu-baseline --no-f64 --concurrent 0 --m2-frames 0 > $OUT/rocprof_m1.json 2> /dev/null
python $R/scripts/prof_summary.py $OUT/prof_m1/m1_results.db $OUT/rocprof_stats_m1.txt > /dev/null
cd $R
grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3; tail -2 $OUT/smoke.log; tail -1 $OUT/bench.err; tail -1 $OUT/check_normals.log; head -8 $OUT/normals_stats.txt; head -14 $OUT/rocprof_stats.txt | cut -c1-80,100-170
