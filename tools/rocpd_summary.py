"""rocprofv3 (ROCm 7.2) writes its results as a rocpd SQLite database; this turns one into the summaries kept under profiles/:
   kernel statistics in the column layout of rocprofv3's own *_kernel_stats.csv (from the kernel-dispatch records of a
   `--kernel-trace --stats` run), and, for a `--pmc` run, the per-kernel mean of each counter.
usage: rocpd_summary.py stats RESULTS.db OUT.csv | pmc RESULTS.db OUT.csv"""
import csv
import json
import math
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from beyond_dof_amd._lib import build_id  # noqa: E402


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select name, start, end from kernels').fetchall()
    acc = {}
    for name, s, e in rows:
        acc.setdefault(name, []).append(e - s)
    total = float(sum(sum(v) for v in acc.values()))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs', 'StdDev'])
        for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            mean = sum(v) / float(len(v))
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / max(len(v) - 1, 1))
            w.writerow([name, len(v), sum(v), round(mean, 6), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 6)])
    json.dump(build_id(), open(os.path.splitext(out)[0] + '.build.json', 'w'), indent=1, sort_keys=True)     # what was profiled


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    cols = [c[1] for c in cur.execute('pragma table_info(pmc_events)')]
    rows = cur.execute('select name, counter_name, counter_value from pmc_events').fetchall()
    acc = {}
    for name, cn, val in rows:
        acc.setdefault((name, cn), []).append(float(val))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(['Kernel_Name', 'Counter_Name', 'Dispatches', 'Mean_Counter_Value', 'Sum_Counter_Value'])
        for (name, cn), v in sorted(acc.items()):
            w.writerow([name, cn, len(v), sum(v) / len(v), sum(v)])
    return cols


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
