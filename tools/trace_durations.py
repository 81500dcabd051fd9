# per-dispatch durations from a rocprofv3 --kernel-trace csv:  python3 tools/trace_durations.py <kernel_trace.csv> [substr ...]
import csv, sys
subs = sys.argv[2:] or ['']
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name']
    if any(s in n for s in subs):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        short = n.replace('void lsqr::', '').split('(')[0][:60]
        print('%10.1f us  +%8.1f us  %s  grid %s wg %s' % ((s - t0) / 1e3, (e - s) / 1e3, short, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))))
