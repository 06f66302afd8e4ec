import csv,sys,glob,collections
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        cnt[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        if 'hive' in k:
            print(k)
            for c,val in v.items(): print('   ',c, val/cnt[(k,c)], 'per dispatch over', cnt[(k,c)])
