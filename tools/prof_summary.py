#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) of bench.py into markdown:
whole-run per-kernel stats (what `--stats` prints) and the breakdown of ONE steady-state step (kernels between
the last two launches of the K1 kernel), which excludes first-call MIOpen/hipBLASLt solver searches.

    python tools/prof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_bench_kernel_trace.md
"""
import sqlite3
import sys


def main(path, marker="rba_reduce"):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                          "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 kernel trace summary ({path.split('/')[-1]})\n")
    print(f"whole run: {sum(r[1] for r in rows)} dispatches, {tot / 1e6:.2f} ms of kernel time\n")
    print("| % | calls | total ms | avg us | min us | max us | kernel |\n|---|---|---|---|---|---|---|")
    for r in rows[:25]:
        print(f"| {r[2] / tot * 100:.1f} | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | `{r[0][:90]}` |")
    k1 = list(c.execute(f"select start, end from kernels where name like '%{marker}%' order by start"))
    if len(k1) >= 2:
        t0, t1 = k1[-2][1], k1[-1][1]
        rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels where start>=? and end<=? "
                              "group by name order by 3 desc", (t0, t1)))
        busy = sum(r[2] for r in rows)
        print(f"\n## one steady-state step (between the last two `{marker}` launches)\n")
        print(f"span {(t1 - t0) / 1e6:.2f} ms, {sum(r[1] for r in rows)} dispatches, {busy / 1e6:.2f} ms busy\n")
        print("| % of busy | calls | total us | avg us | kernel |\n|---|---|---|---|---|")
        for r in rows[:40]:
            print(f"| {r[2] / busy * 100:.1f} | {r[1]} | {r[2] / 1e3:.1f} | {r[3] / 1e3:.1f} | `{r[0][:90]}` |")
        d = [e - s for s, e in k1]
        print(f"\n`{marker}` launches: {len(d)}, avg {sum(d) / len(d) / 1e3:.1f} us, min {min(d) / 1e3:.1f} us, max {max(d) / 1e3:.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
