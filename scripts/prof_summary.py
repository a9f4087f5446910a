"""Dump a rocprofv3 rocpd database (kernel trace) as the plain-text --stats summary kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    lines = ["# rocprofv3 --kernel-trace --stats summary (from %s)" % db_path,
             "%-110s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-110s %8d %14.1f %12.2f %8.2f" % (name[:110], calls, total, avg, pct))
    lines.append("")
    lines.append("# per-dispatch detail of the ICP pass kernels (first 24 dispatches): duration_us grid wg vgpr sgpr lds scratch")
    q = ("select name,duration,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,scratch_size from kernels "
         "where name like '%icp_%' order by start limit 24")
    for r in cur.execute(q):
        lines.append("%-60s %9.1f %8d %5d %4d %4d %6d %4d" % (r[0][:60], r[1] / 1000.0, r[2], r[3], r[4], r[5], r[6], r[7]))
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
