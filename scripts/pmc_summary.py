"""Summarise a rocprofv3 --pmc run (--output-format csv) into a small per-kernel CSV of counter means for profiles/.
usage: python scripts/pmc_summary.py <dir with *_counter_collection.csv [+ *_kernel_trace.csv]> <out.csv> [kernel-name filter]
Launches of one kernel with different grid sizes (e.g. one cloud vs 16 clouds) are reported as separate rows: kernel@grid."""
import csv
import glob
import os
import sys
from collections import defaultdict

def short(k):
    """'void (anonymous namespace)::kern<2, 8, true>((anonymous namespace)::Args)' -> 'kern<2, 8, true>'"""
    k = k.replace('(anonymous namespace)::', '').replace('void ', '')
    depth = 0
    for i, ch in enumerate(k):
        depth += ch == '<'; depth -= ch == '>'
        if ch == '(' and depth == 0:
            return k[:i]
    return k


d, out = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ''
vals = defaultdict(list)
for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r.get('Kernel_Name') or r.get('kernel_name')
            if flt and flt not in k:
                continue
            key = (r.get('Dispatch_Id') or r.get('dispatch_id'), r.get('Counter_Name') or r.get('counter_name'))
            per_dispatch[key] += float(r.get('Counter_Value') or r.get('counter_value'))
            names[key[0]] = k + '@' + str(r.get('Grid_Size') or r.get('grid_size') or '')
    for (disp, cname), v in per_dispatch.items():
        nm, grid = names[disp].rsplit('@', 1)
        vals[(short(nm) + ('@grid=' + grid if grid else ''), cname)].append(v)
for path in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r.get('Kernel_Name')
            if flt and flt not in k:
                continue
            grid = r.get('Grid_Size') or r.get('Grid_Size_X') or ''
            vals[(short(k) + ('@grid=' + str(grid) if grid else ''), 'duration_ns')].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
with open(out, 'w') as f:
    f.write('kernel,counter,mean,launches\n')
    for (k, c), v in sorted(vals.items()):
        v = v[len(v) // 3:] if len(v) >= 6 else v          # drop warm-up launches
        f.write(f'"{k[:120]}",{c},{sum(v) / len(v):.0f},{len(v)}\n')
print(open(out).read()[:6000])
