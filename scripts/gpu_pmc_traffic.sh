#!/bin/bash
# Counter-based traffic, calibrated.  One --pmc set per rocprofv3 run, kernel-trace only.
#   1. scripts/build/pmc_calib (known byte counts: stream / aligned runs / single records / ICP-like gathers / stores; 32 MiB, 128 MiB, 1 GiB)
#   2. bench.py M1 only (icp_fused_kernel: configs[1] and the 8 M-point map) and a few frames of the configs[2] stream
# SETS (env): space-separated names of the counter sets to run (default "ea dram wr"); round 3's first visit ran "fetch write ea hit".
# Output: gpurun_out/pmc_traffic/{calib,m1,stream}_<set>/..., summary gpurun_out/pmc_traffic.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_traffic; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
[ -x $R/scripts/build/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $R/scripts/build/pmc_calib $R/scripts/pmc_calib.hip
declare -A SET
SET[fetch]="FETCH_SIZE"
SET[write]="WRITE_SIZE"
SET[ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
SET[hit]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
SET[dram]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum"
SET[wr]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum"
for name in ${SETS-dram wr}; do
  [ -n "$SKIP_CALIB" ] || { ( cd $R && timeout 120 rocprofv3 --kernel-trace --pmc ${SET[$name]} --output-format csv -d $OUT/calib_$name -o pmc -- scripts/build/pmc_calib > $OUT/calib_$name.log 2>&1 ); echo "calib $name rc=$?"; }
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc ${SET[$name]} --output-format csv -d $OUT/m1_$name -o pmc -- python bench.py --steps 20 --warmup 3 --m2-frames 0 --no-cpu-baseline --concurrent 0 --no-host-seam --no-gicp > $OUT/m1_$name.log 2>&1 ); echo "m1 $name rc=$?"
done
for name in ${STREAM_SETS-dram wr}; do
  ( cd $R && timeout 240 rocprofv3 --kernel-trace --pmc ${SET[$name]} --output-format csv -d $OUT/stream_$name -o pmc -- python scripts/stream_few_frames.py 3 > $OUT/stream_$name.log 2>&1 ); echo "stream $name rc=$?"
done
( cd $R && python scripts/pmc_traffic_summary.py $OUT --json > $R/gpurun_out/pmc_traffic.txt 2>&1 ); echo "summary rc=$?"
( cd $R && python scripts/make_traffic_json.py gpurun_out/pmc_traffic.json > $R/gpurun_out/pmc_traffic_json.log 2>&1 ); echo "json rc=$?"; tail -5 $R/gpurun_out/pmc_traffic_json.log
find $OUT -name "*.csv" -size +8M -delete
