#!/usr/bin/env python3
"""Reads a rocprofv3 rocpd database (--kernel-trace [--memory-copy-trace]) and prints the device timeline of the
long kernels and the large copies: start, duration, gap to the previous long kernel, and how much of each copy ran
under a kernel.  usage: tools/trace_timeline.py <results.db> [min_kernel_ms] [min_copy_bytes]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    min_k = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
    min_c = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
    c = db.cursor()
    names = {}
    try:
        for kid, name in c.execute("select id, kernel_name from rocpd_info_kernel_symbol"):
            names[kid] = name
    except sqlite3.Error:
        pass
    ks = c.execute("select start, end, kernel_id, stream_id from rocpd_kernel_dispatch order by start").fetchall()
    cs = []
    try:
        cs = c.execute("select start, end, size, stream_id from rocpd_memory_copy order by start").fetchall()
    except sqlite3.Error:
        pass
    t0 = min([k[0] for k in ks] + [x[0] for x in cs])
    ev = [("K", *k) for k in ks if k[1] - k[0] >= min_k] + [("C", *x) for x in cs if x[2] >= min_c]
    ev.sort(key=lambda e: e[1])
    last_end = None
    for e in ev:
        if e[0] == "K":
            gap = (e[1] - last_end) / 1e6 if last_end is not None else 0.0
            print("K %10.3f ms  dur %8.3f  gap %7.3f  stream %s  %s" % ((e[1] - t0) / 1e6, (e[2] - e[1]) / 1e6, gap, e[4],
                                                                    names.get(e[3], e[3])[:60]))
            last_end = e[2] if last_end is None else max(last_end, e[2])
        else:
            under = sum(max(0, min(e[2], k[1]) - max(e[1], k[0])) for k in ks if k[1] - k[0] >= min_k)
            print("C %10.3f ms  dur %8.3f  %6.1f MB  %5.1f GB/s  under kernels %3.0f %%  stream %s"
                  % ((e[1] - t0) / 1e6, (e[2] - e[1]) / 1e6, e[3] / 1e6, e[3] / (e[2] - e[1]), 100.0 * under / (e[2] - e[1]), e[4]))


if __name__ == "__main__":
    main()
