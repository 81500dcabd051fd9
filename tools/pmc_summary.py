# summarise a rocprofv3 --pmc counter_collection.csv: per kernel dispatch, counters side by side
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
d = collections.OrderedDict()
for r in rows:
    if not any(t in r['Kernel_Name'] for t in ('k_scan','k_syrk','k_solve_dense','k_estimate_dense','k_reduce')): continue
    key = (int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0][-40:], r['VGPR_Count'], r['SGPR_Count'])
    d.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
for k, v in d.items():
    print(k, {n: ('%.3g' % x) for n, x in v.items()})
