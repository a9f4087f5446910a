"""One lidar frame of the stream as the GPU saw it: every dispatch between two consecutive frame ingests (pack_strided_f32_kernel), with its
start (us after the frame's first dispatch), duration and the idle gap in front of it.  From a rocprofv3 --kernel-trace rocpd database."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"o3ds::", "", n)
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    return n.split("(")[0][:70]


def main(db_path, out_path, frame=60):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name,start,end from kernels order by start"))
    marks = [k for k, r in enumerate(rows) if "pack_strided_f32_kernel" in r[0]]
    a, b = marks[frame], marks[frame + 1]
    t0 = rows[a][1]
    lines = ["# dispatches of stream frame %d (between two ingests): start_us dur_us gap_us kernel" % frame]
    prev_end = None
    busy = 0.0
    for name, s, e in rows[a:b]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        lines.append("%9.1f %8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(name)))
        prev_end = e
        busy += (e - s) / 1e3
    span = (rows[b][1] - t0) / 1e3
    lines.append("# %d dispatches, %.1f us busy of %.1f us between the two ingests" % (b - a, busy, span))
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
