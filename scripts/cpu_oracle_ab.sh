#!/bin/bash
# CPU baseline of configs[1]: the oracle's correspondence loop dealt out statically (as before) against dynamically (now), per thread count
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp -fPIC -std=c11 -shared "-DORC_SEARCH_SCHEDULE=schedule(static)" -o /tmp/libo3d_oracle_static.so oracle/o3d_oracle.c -lm
for v in dynamic static; do
  if [ $v = static ]; then export O3DS_ORACLE_LIB=/tmp/libo3d_oracle_static.so; else unset O3DS_ORACLE_LIB; fi
  for bind in close spread; do
    echo "== search loop $v, OMP_PROC_BIND=$bind OMP_PLACES=cores"
    OMP_PROC_BIND=$bind OMP_PLACES=cores timeout 300 python scripts/cpu_scaling.py 2>&1 | grep "threads\|nproc"
  done
done > $OUT/cpu_oracle_ab.txt 2>&1
cat $OUT/cpu_oracle_ab.txt
