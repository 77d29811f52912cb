"""Kernel statistics (calls, total, average, share) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats` writes <dir>/<name>_results.db in ROCm 7.2) as CSV.
usage: python scripts/rocpd_stats.py run_results.db [out.csv] [steps]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                  'from kernels group by name order by sum(duration) desc').fetchall()
tot = float(sum(r[2] for r in rows))
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
out = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 and sys.argv[2] != '-' else sys.stdout)
out.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'] +
             (['MsPerStep'] if steps else []))
for n, c, s, a, mn, mx in rows:
    out.writerow([n, c, s, '%.1f' % a, '%.2f' % (100.0 * s / tot), mn, mx] +
                 (['%.3f' % (s / 1e6 / steps)] if steps else []))
