#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (the default output format of rocprofv3 --kernel-trace).
usage: python tools/kernel_table.py <results.db> [steps]   -> name, launches, average us, share of kernel time"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    rows = db.execute("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from rocpd_kernel_dispatch d "
                      "join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by 1 order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    print(f"total kernel time {tot / 1e6:.2f} ms" + (f" = {tot / 1e6 / steps:.3f} ms per step" if steps else ""))
    for n, c, a, s in rows:
        print(f"{s / tot * 100:5.1f}% {c:6d} {a / 1e3:9.1f} us  {n[:120]}")


if __name__ == "__main__":
    main()
