#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) of bench.py into markdown:
whole-run per-kernel stats (what `--stats` prints) and the breakdown of ONE steady-state step (kernels between
the last two launches of the K1 kernel), which excludes first-call MIOpen/hipBLASLt solver searches.

    python tools/prof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_bench_kernel_trace.md
    python tools/prof_summary.py --sequence gpurun_out/prof/bench_results.db      # every dispatch of that step, in launch order, with its grid
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
    shown = rows[:25] + [r for r in rows[25:] if marker in r[0]]          # the roofline kernel's row is always there, wherever it ranks
    for r in shown:
        print(f"| {r[2] / tot * 100:.1f} | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.1f} | {r[4] / 1e3:.1f} | {r[5] / 1e3:.1f} | `{r[0][:90]}` |")
    if len(rows) > len(shown):
        rest = rows[len(shown):] if len(shown) == 25 else [r for r in rows[25:] if marker not in r[0]]
        print(f"| {sum(r[2] for r in rest) / tot * 100:.1f} | {sum(r[1] for r in rest)} | {sum(r[2] for r in rest) / 1e6:.3f} | | | | ({len(rest)} more kernels) |")
    # bench.py's `roofline.kernel` (and every other kernel of that family), one row EACH -- two kernels are never merged under one label
    fam = [r for r in rows if marker in r[0]]
    if fam:
        print(f"\n## `{marker}*` kernels (bench.py `roofline.kernel` is one of them), whole run\n")
        print("| calls | avg us | min us | max us | kernel |\n|---|---|---|---|---|")
        for r in fam:
            print(f"| {r[1]} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | `{r[0][:120]}` |")
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
        print(f"\n(step boundaries: the {len(k1)} launches of any `{marker}*` kernel; per-kernel durations are in the table above)")


def sequence(path, marker="rba_reduce"):
    """one steady-state step (single-stream run) dispatch by dispatch: the per-stage view -- the same instantiation serves several stages"""
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = next((n for n in ("grid_x", "grid_size_x", "grid_size") if n in cols), None)
    wx = next((n for n in ("workgroup_x", "workgroup_size_x", "workgroup_size") if n in cols), None)
    k1 = list(c.execute(f"select start, end from kernels where name like '%{marker}%' order by start"))
    t0, t1 = k1[-2][1], k1[-1][1]
    sel = "name, start, end" + (f", {gx}" if gx else "") + (f", {wx}" if wx else "")
    rows = list(c.execute(f"select {sel} from kernels where start>=? and end<=? order by start", (t0, t1)))
    print(f"# one steady-state step, {len(rows)} dispatches in launch order (span {(t1 - t0) / 1e3:.0f} us)\n")
    print("| # | start us | us | gap before us | workgroups | threads | kernel |\n|---|---:|---:|---:|---:|---:|---|")
    prev = t0
    for i, r in enumerate(rows):
        g = r[3] if gx else 0
        w = r[4] if (gx and wx) else (r[3] if wx else 0)
        nm = r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        print(f"| {i} | {(r[1] - t0) / 1e3:.1f} | {(r[2] - r[1]) / 1e3:.1f} | {(r[1] - prev) / 1e3:.1f} | {g // w if w else g} | {w} | `{nm[:110]}` |")
        prev = r[2]


if __name__ == "__main__":
    if sys.argv[1] == "--sequence":
        sequence(*sys.argv[2:])
    else:
        main(*sys.argv[1:])
