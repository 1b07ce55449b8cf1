import csv, glob, collections, sys
for tag in sys.argv[1:]:
    fs = glob.glob("gpurun_out/%s_pmc/*/*_counter_collection.csv" % tag)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "prd::k_trace" not in k and "prd::k_shade" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        print(tag, k, "launches", len(n[k]))
        for c, v in sorted(agg[k].items()):
            print("    %-34s %.4g per launch" % (c, v / len(n[k])))
