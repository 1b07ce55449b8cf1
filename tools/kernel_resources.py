#!/usr/bin/env python3
"""Register / LDS / scratch budget of every kernel in the SHIPPED code objects (pearray_amd/csrc/build/*.o), read from their
AMDGPU metadata notes with llvm-readelf -- not from a profiler.  usage: python tools/kernel_resources.py [out.json]"""
import glob, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
out = {}
for obj in sorted(glob.glob(os.path.join(ROOT, "pearray_amd", "csrc", "build", "*.o"))):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        os.symlink(obj, tmp)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in glob.glob(tmp + ".*") if "gfx950" in f]
        if not co:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co[0]], capture_output=True, text=True).stdout
    for block in notes.split("- .agpr_count:")[1:]:
        get = lambda key: (re.search(r"\." + key + r":\s+(\S+)", block) or [None, None])[1]  # noqa: E731
        name = get("name")
        if not name:
            continue
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        if "k_path_persistent" not in demangled and "k_path_latency" not in demangled and "k_trace" not in demangled and "k_service" not in demangled and "k_shade" not in demangled:
            continue
        out[demangled + " [" + os.path.basename(obj) + "]"] = {
            "vgpr_count": int(get("vgpr_count")), "agpr_count": int(block.split()[0]), "sgpr_count": int(get("sgpr_count")),
            "vgpr_spill_count": int(get("vgpr_spill_count")), "sgpr_spill_count": int(get("sgpr_spill_count")),
            "scratch_bytes_per_lane": int(get("private_segment_fixed_size")), "lds_bytes_per_block": int(get("group_segment_fixed_size")),
            "max_flat_workgroup_size": int(get("max_flat_workgroup_size"))}
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_kernel_resources.json")
json.dump({"source": "llvm-readelf --notes of the gfx950 code objects in pearray_amd/csrc/build (tools/kernel_resources.py)", "kernels": out}, open(dst, "w"), indent=1)
for k, v in out.items():
    print("%-70s VGPR %3d spills %4d SGPR spills %4d scratch %4d B LDS %6d B" % (k[:70], v["vgpr_count"], v["vgpr_spill_count"], v["sgpr_spill_count"], v["scratch_bytes_per_lane"], v["lds_bytes_per_block"]))
