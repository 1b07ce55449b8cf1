"""Summarise a tools/gpu_profile_quick.sh run: per-depth kernel times of the last iteration + PMC per launch."""
import csv, glob, collections, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "q"
base = "gpurun_out/%s_" % tag
f = glob.glob(base + "trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
rg = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]]
last = rows[rg[-1]:]
print("last iteration:")
for r in last:
    n = r["Kernel_Name"].split("(")[0].replace("void prd::", "").replace("prd::", "")
    if n.startswith("k_"):
        print("  %-28s grid %8s  %.3f ms  vgpr %s lds %s" % (n, r["Grid_Size_X"], d(r), r["VGPR_Count"], r["LDS_Block_Size"]))
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
busy = sum(d(r) for r in last)
print("  span %.3f ms, kernel busy %.3f ms" % (span, busy))
out = {}
for sub in ("sq", "l2", "fetch", "write"):
    fs = glob.glob(base + sub + "/*/*_counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "prd::k_" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        out.setdefault(k, {})
        for c, v in agg[k].items():
            out[k][c] = v / len(n[k])
for k, v in out.items():
    print(k)
    print("   ", {a: "%.4g" % b for a, b in v.items()})
    if "SQ_WAVE_CYCLES" in v:
        print("    wait_any %.2f  wait_inst %.2f  active %.2f ; VALU/VMEM %.1f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["SQ_INSTS_VALU"] / max(v["SQ_INSTS_VMEM"], 1)))
    if "TCC_HIT_sum" in v:
        print("    L2 hit rate %.3f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
json.dump(out, open("gpurun_out/%s_pmc.json" % tag, "w"), indent=1)
