"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats) into a small CSV for profiles/."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
with open(out, 'w') as f:
    f.write('kernel,calls,total_us,avg_us,percent\n')
    for name, calls, tot, avg, pct in rows:
        name = name.replace('"', "'")
        if len(name) > 140:
            name = name[:137] + '...'
        f.write(f'"{name}",{calls},{tot / 1e3 if tot > 1e7 else tot:.3f},{avg / 1e3 if tot > 1e7 else avg:.3f},{pct:.3f}\n')
print(open(out).read()[:3000])
