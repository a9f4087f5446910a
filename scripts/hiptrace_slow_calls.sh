#!/bin/bash
# HIP API trace of the config-2 stream: the slowest individual runtime calls of the steady frames, with what ran just before them
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp; rm -rf $OUT/prof_hip
timeout 300 rocprofv3 --hip-trace --kernel-trace -d $OUT/prof_hip -o s -- python $R/scripts/bench_stream.py --frames 12 --cpu-frames 0 > /dev/null 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect("$OUT/prof_hip/s_results.db"); cur=db.cursor()
rows=list(cur.execute("select name, start, end from regions order by start"))
t_end=rows[-1][2]
# steady part: the last 40 % of the trace
t0=rows[0][1]; cut=t0+(t_end-t0)*0.6
steady=[(n,s,e) for n,s,e in rows if s>=cut]
print("steady api calls", len(steady), "span ms", (t_end-cut)/1e6)
slow=sorted(steady, key=lambda r:-(r[2]-r[1]))[:25]
idx={id(r):i for i,r in enumerate(steady)}
for r in sorted(slow, key=lambda r:r[1]):
    i=steady.index(r)
    prev=[steady[j][0] for j in range(max(0,i-3),i)]
    print("%9.1f us  %-28s at +%8.1f us   after: %s"%((r[2]-r[1])/1e3, r[0][:28], (r[1]-cut)/1e3, " > ".join(p[:22] for p in prev)))
PY
rm -rf $OUT/prof_hip
