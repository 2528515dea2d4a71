import csv, collections, glob, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc"
for f in sorted(glob.glob(f"gpurun_out/{tag}?/p_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "gemm" not in k and "attn" not in k: continue
        print(k, {c: f"{sum(x)/len(x):.3e}" for c, x in v.items()})
