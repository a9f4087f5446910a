cd ${GRAFT_REPO_ROOT:-.}; R=$(pwd); mkdir -p gpurun_out; python scripts/carve_profile.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_carve && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_carve -o c -- python $R/scripts/carve_profile.py > /dev/null 2>&1
python $R/scripts/prof_summary.py $R/gpurun_out/prof_carve/c_results.db $R/gpurun_out/rocprof_stats_carve.txt > /dev/null; rm -f $R/gpurun_out/prof_carve/*.db; head -30 $R/gpurun_out/rocprof_stats_carve.txt | cut -c1-90,100-170
