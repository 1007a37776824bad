"""Kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite database.
usage: python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for n, c, s, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, c, s, a, 100.0 * s / tot, mn, mx))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    for n, c, s, a, mn, mx in rows[:40]:
        short = re.sub(r"\(.*", "", n)[-70:]
        print("%-70s %6d %10.3f ms %9.1f us %6.2f%%" % (short, c, s / 1e6, a / 1e3, 100.0 * s / tot))
    print("total kernel time %.2f ms" % (tot / 1e6))


if __name__ == "__main__":
    main()
