"""Summarise a rocprofv3 rocpd database (kernel-trace): per-kernel totals, like --stats.
usage: python tools_prof_summary.py results.db [top_n] [steps]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f'"Name","Calls","TotalDurationMs","AverageUs","MinUs","MaxUs","Percentage"')
    for n, k, ms, us, mn, mx in rows[:top]:
        print(f'"{n[:160]}",{k},{ms:.3f},{us:.1f},{mn:.1f},{mx:.1f},{100 * ms / tot:.2f}')
    print(f'"TOTAL",{sum(r[1] for r in rows)},{tot:.3f},,,,100')


if __name__ == '__main__':
    main()
