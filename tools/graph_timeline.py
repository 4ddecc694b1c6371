#!/usr/bin/env python
"""Timeline statistics of a captured step from a rocprofv3 --kernel-trace database (rocpd sqlite): per replay the wall
time, the summed kernel time, how much of the wall had 0 / 1 / 2 / 3+ kernels running, and the kernels sorted by summed
time.  usage: graph_timeline.py results.db <name of the step's first kernel> [replays to skip]"""
import collections
import sqlite3
import sys


def main():
    db, first = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,start,end,queue_id from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    steps = [rows[idx[i]:idx[i + 1]] for i in range(skip, len(idx) - 1)]
    print(f"{len(idx)} replays found, {len(steps)} analysed; {len(steps[0])} launches per step, "
          f"{len(set(r[3] for r in steps[0]))} hardware queues")
    walls, busys, conc = [], [], collections.Counter()
    tot = collections.defaultdict(lambda: [0, 0.0])
    for seg in steps:
        t0, t1 = seg[0][1], max(r[2] for r in seg)
        walls.append((t1 - t0) / 1e3)
        busys.append(sum(r[2] - r[1] for r in seg) / 1e3)
        ev = sorted([(r[1], 1) for r in seg] + [(r[2], -1) for r in seg])
        n, last = 0, t0
        for t, d in ev:
            conc[min(n, 3)] += (t - last) / 1e3
            n, last = n + d, t
        for name, s, e, q in seg:
            k = name.split("(")[0][:64]
            tot[k][0] += 1
            tot[k][1] += (e - s) / 1e3
    ns = len(steps)
    print(f"wall {sum(walls) / ns:8.1f} us/step   summed kernel time {sum(busys) / ns:8.1f} us/step")
    print("wall with n kernels running: " + "  ".join(f"{k}{'+' if k == 3 else ''}: {v / ns:7.1f} us" for k, v in sorted(conc.items())))
    for k, (cnt, t) in sorted(tot.items(), key=lambda x: -x[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
        print(f"{cnt / ns:7.1f} x {t / cnt:7.1f} us = {t / ns:8.1f} us  {k}")


if __name__ == "__main__":
    main()
