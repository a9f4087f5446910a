#!/bin/bash
# Counter-based traffic, calibrated.  One --pmc set per rocprofv3 run, kernel-trace only.
#   1. scripts/build/pmc_calib (known byte counts: stream / aligned runs / single records / ICP-like gathers / stores; 32 MiB, 128 MiB, 1 GiB)
#   2. bench.py M1 only (icp_fused_kernel) and the configs[2] stream (normals, scatter, merge kernels)
# Output: gpurun_out/pmc_traffic/{calib,m1,stream}_<set>/..., summary gpurun_out/pmc_traffic.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_traffic; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|WRITE_SIZE|TCC_EA0?_(RD|WR)REQ|TCC_(HIT|MISS|REQ)|TCC_EA0?_RD_UNCACHED|MALL|TCC_BUBBLE" | head -80 > $OUT/counters_available.txt
[ -x $R/scripts/build/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $R/scripts/build/pmc_calib $R/scripts/pmc_calib.hip
( cd $R && timeout 120 scripts/build/pmc_calib > $OUT/calib_plain.txt 2>&1 ); echo "calib plain rc=$?"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum")
NAMES=(fetch write ea hit)
for i in 0 1 2 3; do
  ( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc ${SETS[$i]} --output-format csv -d $OUT/calib_${NAMES[$i]} -o pmc -- scripts/build/pmc_calib > $OUT/calib_${NAMES[$i]}.log 2>&1 ); echo "calib ${NAMES[$i]} rc=$?"
done
for i in 0 1 2; do
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc ${SETS[$i]} --output-format csv -d $OUT/m1_${NAMES[$i]} -o pmc -- python bench.py --steps 20 --warmup 3 --m2-frames 0 --no-cpu-baseline --concurrent 0 > $OUT/m1_${NAMES[$i]}.log 2>&1 ); echo "m1 ${NAMES[$i]} rc=$?"
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc ${SETS[$i]} --output-format csv -d $OUT/stream_${NAMES[$i]} -o pmc -- python scripts/bench_stream.py --frames 40 > $OUT/stream_${NAMES[$i]}.log 2>&1 ); echo "stream ${NAMES[$i]} rc=$?"
done
( cd $R && python scripts/pmc_traffic_summary.py $OUT > $R/gpurun_out/pmc_traffic.txt 2>&1 ); echo "summary rc=$?"
# keep the per-dispatch CSVs small enough to travel back
find $OUT -name "*.csv" -size +8M -delete
