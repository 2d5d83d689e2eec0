import csv, sys
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Name']:
            print(f.split('/')[2], r['Name'][13:60], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(int(r['MinNs'])/1e3,1), 'max', round(int(r['MaxNs'])/1e3,1))
