#!/bin/bash
# HIP API statistics of the configs[2] stream alone (100 frames): how many synchronisations, copies and launches a frame costs on the host side
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_hip
timeout 400 rocprofv3 --hip-trace --stats -d $OUT/prof_hip -o s -- python $R/scripts/bench_stream.py --frames 100 > $OUT/stream_hip.json 2>/dev/null
ls $OUT/prof_hip | head
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/prof_hip/*.db")[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(duration) / 1e3, avg(duration) / 1e3 from regions where category like 'HIP_RUNTIME%' group by name order by 3 desc"))
with open("$OUT/hip_api_stats_stream.txt", "w") as out:
    out.write("# HIP runtime API calls of scripts/bench_stream.py --frames 100 (rocprofv3 --hip-trace): name, calls, total us, average us\n")
    for r in rows:
        out.write("%-44s %8d %12.1f %9.2f\n" % r)
print(open("$OUT/hip_api_stats_stream.txt").read()[:5000])
PY
rm -f $OUT/prof_hip/*.db
