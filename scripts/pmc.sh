#!/bin/bash
# PMC counter passes for the ICP pass kernel (each --pmc set in its own run; no trace domains beyond kernel-trace).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -o -E "\b(SQ_[A-Z_0-9]+|TCC_[A-Za-z_0-9]+|TCP_[A-Za-z_0-9]+|GRBM_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|L2CacheHit|MemUnitStalled|VALUBusy|OccupancyPercent)\b" $OUT/counters_list.txt | sort -u > $OUT/counter_names.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --m2-frames 0 --concurrent 0 > $OUT/p$i.log 2>&1
  echo "set $i ($set) rc=$?"
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-60:]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            if "icp_" not in k: continue
            print(k)
            for c, v in cs.items():
                v2 = v[1:] if len(v) > 1 else v   # drop the first (misaligned) pass
                print("   %-32s n=%4d mean=%14.1f  first=%14.1f" % (c, len(v), sum(v2)/len(v2), v[0]))
PY
