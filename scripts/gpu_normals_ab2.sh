#!/bin/bash
# normals kernels under rocprofv3: CASES="label|lib|ENV=1 ENV2=2;..." (lib relative to the repo), kernel averages per case, twice; CHECK=lib runs check_normals.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
IFS=';' read -ra CS <<< "$CASES"
for rep in 1 2; do
for c in "${CS[@]}"; do
  IFS='|' read -r label lib envs <<< "$c"
  rm -rf $OUT/prof_nl
  env $envs O3DS_BACKEND_LIB=$R/$lib REPS=20 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_nl -o s -- python $R/scripts/${SCRIPT:-normals_one.py} > $OUT/nl.log 2>/dev/null
  python $R/scripts/prof_summary.py $OUT/prof_nl/s_results.db $OUT/nl_stats.txt > /dev/null
  echo "$label: $(grep normals_kernel $OUT/nl_stats.txt | awk -F"|" "{print}" | grep -o "[0-9.]* *[0-9.]* *[0-9.]* *[0-9.]*$" | tr "\n" ";") | $(cat $OUT/nl.log)"
done
done
cd $R
if [ -n "$CHECK" ]; then O3DS_BACKEND_LIB=$R/$CHECK timeout 300 python scripts/check_normals.py 2>&1 | tail -1; fi
