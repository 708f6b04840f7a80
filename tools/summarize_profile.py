#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory (gpurun_out/prof_<tag>) into
profiles/<round>/<tag>_*.{csv,txt,json} and profiles/traffic_latest.json.
usage: summarize_profile.py gpurun_out/prof_<tag> profiles/r01 <tag>"""
import collections, csv, glob, json, os, shutil, sys
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if "mh_step" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in agg.items()}
with open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w") as o:
    o.write("rocprofv3 --pmc passes (one run per pass) of `python3 bench.py --no-cpu-baseline --steps 16 --warmup 16`;\n"
            "mh_step_kernel, mean per launch (16 MH steps x 65536 chains, 1024 waves)\n")
    for k in sorted(mean):
        o.write(f"{k:24s} launches={len(agg[k])} mean={mean[k]:.6g}\n")
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE is in KB and reports half the bytes of wide
        # coalesced reads on gfx950 (x2); WRITE_SIZE (KB) is exact
        fetch = mean["FETCH_SIZE"] * 1024 * 2
        write = mean["WRITE_SIZE"] * 1024
        o.write(f"HBM-side bytes per launch: fetch {fetch:.4g} (FETCH_SIZE KB x 1024 x 2, gfx950 correction) + write {write:.4g}\n")
        json.dump({"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "tag": tag,
                   "note": "FETCH_SIZE x2 (gfx950 under-count of wide coalesced reads) + WRITE_SIZE, per mh_step_kernel launch of 16 steps x 65536 chains"},
                  open(os.path.join(os.path.dirname(dst.rstrip('/')), "traffic_latest.json"), "w"))
print(open(os.path.join(dst, f"{tag}_pmc_summary.txt")).read())
