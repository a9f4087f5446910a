#!/bin/bash
# instruction-mix PMC passes (one --pmc set per run, kernel-trace only) for an arbitrary command:  scripts/pmc_cmd.sh <label> <cmd...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=$1; shift; OUT=$R/gpurun_out/pmc_$L; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64" \
           "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum"; do
  i=$((i+1))
  ( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- "$@" > $OUT/p$i.log 2>&1 ); echo "set $i rc=$?"
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
            if "icp_fused" not in k and "normals" not in k: continue
            for c, v in cs.items():
                v = sorted(v)
                print("%-58s %-30s n=%4d median=%14.1f max=%14.1f" % (k[-58:], c, len(v), v[len(v)//2], v[-1]))
PY
