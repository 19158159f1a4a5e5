import csv,glob,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k)
    for c,x in v.items(): print('   ',c,len(x),sum(x)/len(x))
