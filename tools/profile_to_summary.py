"""Turn a tools/gpu_profile.sh run (gpurun_out/TAG_*) into the committed artefacts profiles/TAG_kernel_stats.csv and
profiles/TAG_pmc_summary.json.  usage: profile_to_summary.py TAG ITERS_PMC   (ITERS_PMC = iterations rendered in each --pmc pass)"""
import collections, csv, glob, json, os, shutil, sys
tag, iters = sys.argv[1], int(sys.argv[2])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", tag + "_")
newest = lambda files: sorted(files, key=os.path.getmtime)[-1:]   # noqa: E731  -- gpurun_out keeps the files of earlier runs
stats = newest(glob.glob(base + "trace/*/*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(root, "profiles", tag + "_kernel_stats.csv"))
out = collections.OrderedDict()
for sub in ("fetch", "write", "l2", "sq", "ta", "tcp"):
    fs = newest(glob.glob(base + sub + "/*/*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "prd::k_" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in agg:
        o = out.setdefault(k, collections.OrderedDict())
        for c, v in sorted(agg[k].items()):
            o[c + "_per_launch"] = v / len(n[k])
            o[c + "_per_iteration"] = v / iters
        o["launches_" + sub] = len(n[k])
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_sha16: bench.py quotes a summary only for the build it was taken from)
json.dump({"kernel_source_sha16": bench.kernel_source_sha16(), "command": "rocprofv3 --pmc <counter set> --kernel-trace -- python3 bench.py --steps %d --warmup 1 --profile-only "
                      "(one pass per counter set; tools/gpu_profile.sh; the warm-up launch runs the instrumented variant, the plain kernel the %d steps)" % (iters, iters),
           "iterations_per_pass": iters,
           "note": "FETCH_SIZE/WRITE_SIZE in KiB as reported by rocprofv3; FETCH_SIZE under-reports wide reads by 2x on gfx950 "
                   "(MI355X_MICROARCH.md, HBM); *_per_iteration = total over the pass / iterations rendered (raygen-to-fold of every pixel once)",
           "kernels": out}, open(os.path.join(root, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    line = k
    if "FETCH_SIZE_per_iteration" in v:
        line += "  HBM/iter %.3f GB" % ((2 * v["FETCH_SIZE_per_iteration"] + v.get("WRITE_SIZE_per_iteration", 0)) * 1024 / 1e9)
    if "SQ_WAVE_CYCLES_per_launch" in v:
        line += "  active %.2f wait_any %.2f VALU/VMEM %.1f" % (v["SQ_ACTIVE_INST_ANY_per_launch"] / v["SQ_WAVE_CYCLES_per_launch"],
                 v["SQ_WAIT_ANY_per_launch"] / v["SQ_WAVE_CYCLES_per_launch"], v["SQ_INSTS_VALU_per_launch"] / max(v["SQ_INSTS_VMEM_per_launch"], 1))
    if "TCC_HIT_sum_per_launch" in v:
        line += "  L2 hit %.3f" % (v["TCC_HIT_sum_per_launch"] / (v["TCC_HIT_sum_per_launch"] + v["TCC_MISS_sum_per_launch"]))
    print(line)
