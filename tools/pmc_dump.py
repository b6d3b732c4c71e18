import csv, glob, os, sys, collections
KERNEL = os.environ.get("FX_PMC_KERNEL", "k_fasta_comp")
d=sys.argv[1]
rows=collections.defaultdict(dict)
for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL not in r["Kernel_Name"]: continue
        key=(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-30:], r.get("Grid_Size",""))
        rows[key][r["Counter_Name"]]=float(r["Counter_Value"])
for k in sorted(rows, key=lambda x:int(x[0])):
    v=rows[k]
    print(k, {a:round(b) for a,b in v.items()})
