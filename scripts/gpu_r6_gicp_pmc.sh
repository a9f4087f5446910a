#!/bin/bash
# GeneralizedIcp on configs[1]: instruction mix and L2 <-> fabric requests of icp_fused_kernel<P4f, false, 256, 4, true> per launch
# (separate --pmc passes, kernel-trace only), pass by pass for the first registration
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_gicp; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum TCC_HIT_sum"; do
  i=$((i+1))
  ( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python scripts/gicp_one.py > $OUT/p$i.log 2>&1 ); echo "set $i rc=$?"
done
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "icp_fused" not in r["Kernel_Name"]: continue
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
# the two counter sets come from two runs of the same program: dispatch ids repeat, the k-th fused launch of each run is the same launch
print("# GeneralizedIcp, configs[1] (65 536 queries, 1 M-point map), icp_fused_kernel<P4f, false, 256, 4, true>: per launch of the first registration")
print("# pass   VALU/wave  SALU/wave  VMEM/wave  LDS/wave   L2 read req -> fabric   write req   bytes at 64 B/req   x algorithmic (15.73 MB)")
first = [d for d in ids][:12]
for k, d in enumerate(first):
    c = rows[d]; w = c.get("SQ_WAVES", 0) or 1
    rd, wr = c.get("TCC_EA0_RDREQ_sum", float("nan")), c.get("TCC_EA0_WRREQ_sum", float("nan"))
    print("%4d  %9.0f  %9.0f  %9.0f  %8.0f  %14.0f  %12.0f  %14.2f MB  %8.2f" % (k, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_VMEM", 0) / w,
          c.get("SQ_INSTS_LDS", 0) / w, rd, wr, (rd + wr) * 64 / 1e6, (rd + wr) * 64 / (65536 * 240)))
PY
