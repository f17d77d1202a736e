"""Mean PMC counter values per kernel from rocprofv3 --pmc CSV output (first N dispatches of each kernel+grid)."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0][-48:] + ' g' + r['Grid_Size']
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'at::' in k or 'elementwise' in k: continue
    print(k)
    print('   ' + '  '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
