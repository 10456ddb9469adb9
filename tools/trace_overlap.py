#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace database of bench.py: over the steady-state part of the run, wall span, union of kernel intervals (time with
at least one kernel running), sum of kernel durations, average concurrency, and the share of the span with 0 / 1 / 2 / 3+ kernels in flight.
    python tools/trace_overlap.py <results.db> [skip_fraction]"""
import sqlite3
import sys


def main(path, skip=0.5):
    c = sqlite3.connect(path)
    rows = sorted(c.execute("select start, end from kernels"))
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
    rows = [(a, b) for a, b in rows if a >= t_lo]
    ev = sorted([(a, 1) for a, b in rows] + [(b, -1) for a, b in rows])
    hist = {}
    depth, last = 0, ev[0][0]
    for t, d in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        depth += d
        last = t
    span = ev[-1][0] - ev[0][0]
    busy = span - hist.get(0, 0)
    tot = sum(b - a for a, b in rows)
    print(f"steady-state window: span {span / 1e6:.2f} ms, {len(rows)} dispatches; at least one kernel running {busy / span * 100:.1f} % of it; "
          f"sum of kernel durations {tot / 1e6:.2f} ms = {tot / span:.2f}x the span")
    print("share of the span with k kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / span * 100:.1f} %" for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
