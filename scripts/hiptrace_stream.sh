#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/prof_hip
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $OUT/prof_hip -o s -- python $R/scripts/bench_stream.py --frames 20 --cpu-frames 0 > $OUT/hiptrace_stream.json 2>/dev/null
python - <<PY
import sqlite3
db=sqlite3.connect("$OUT/prof_hip/s_results.db"); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
q="select name, count(*), sum(end-start)/1000.0 as us from regions group by name order by us desc limit 25"
try:
    for r in cur.execute(q): print("%-40s %7d %12.1f us"%(r[0][:40], r[1], r[2]))
except Exception as e:
    print("err", e, [t for t in tabs if 'region' in t])
PY
cat $OUT/hiptrace_stream.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['gpu_ms_per_scan'])"
