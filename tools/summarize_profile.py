#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory (gpurun_out/prof_<tag>) into
profiles/<round>/<tag>_*.{csv,txt,json} and profiles/traffic_latest.json.
usage: summarize_profile.py gpurun_out/prof_<tag> profiles/r02 <tag>"""
import collections, csv, glob, json, os, shutil, sys
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_stats.json"), os.path.join(dst, f"{tag}_bench_under_rocprof.json"))
# the stats pass: bench.py's timed region is its last steps/16 launches of mh_step_kernel (the ones before are
# the untimed step-size adaptation and warm-up); report their average beside the all-launch average of the stats csv
bench = json.loads(open(os.path.join(src, "bench_stats.json")).read().strip().splitlines()[-1])
n_timed = max(1, int(bench["roofline"].get("launches", bench["steps"] // 16)))
trace_note = ""
for f in glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "mh_step" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
    full = [d for d, r in zip(dur, rows)]
    trace_note = (f"kernel trace: {len(dur)} mh_step_kernel launches, mean {sum(dur)/len(dur):.3f} ms over all; "
                  f"the {n_timed} timed launches (last in the trace) mean {sum(dur[-n_timed:])/n_timed:.3f} ms; "
                  f"bench.py's own HIP-event figure for them: {bench['roofline']['kernel_ms_per_launch']:.3f} ms\n")
# PMC passes run `--steps 16 --warmup 16`: the last two dispatches of mh_step_kernel are the warm-up and the timed
# launch in the adapted regime; the earlier ones are the adaptation phase
agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mh_step" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in per.items():
        v.sort()
        agg[k] += [x for _, x in v[-2:]]
mean = {k: sum(v) / len(v) for k, v in agg.items()}
with open(os.path.join(dst, f"{tag}_pmc_summary.txt"), "w") as o:
    o.write("rocprofv3 --pmc passes (one run per pass) of `python3 bench.py --no-cpu-baseline --ess-chains 0 ... --steps 16 --warmup 16`;\n"
            f"mh_step_kernel, mean per launch over the last two launches (adapted regime; 16 MH steps x {bench['config']['chains_per_gpu']} chains)\n"
            f"workload: {bench['config']['workload']}\n")
    o.write(f"arithmetic of this run: {bench['dtype']}\n")
    o.write(trace_note)
    for k in sorted(mean):
        o.write(f"{k:24s} launches={len(agg[k])} mean={mean[k]:.6g}\n")
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE is in KB and reports half the bytes of wide
        # coalesced reads on gfx950 (x2); WRITE_SIZE (KB) is exact
        fetch = mean["FETCH_SIZE"] * 1024 * 2
        write = mean["WRITE_SIZE"] * 1024
        o.write(f"HBM-side bytes per launch: fetch {fetch:.4g} (FETCH_SIZE KB x 1024 x 2, gfx950 correction) + write {write:.4g}\n")
        json.dump({"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "tag": tag,
                   "chains": bench["config"]["chains_per_gpu"], "steps_per_launch": 16,
                   "note": f"FETCH_SIZE x2 (gfx950 under-count of wide coalesced reads) + WRITE_SIZE, per mh_step_kernel launch of 16 steps x {bench['config']['chains_per_gpu']} chains"},
                  open(os.path.join(os.path.dirname(dst.rstrip('/')),
                                    "traffic_fast_mode.json" if "fast" in tag else
                                    "traffic_regions.json" if bench["config"].get("regions") else "traffic_latest.json"), "w"))
print(open(os.path.join(dst, f"{tag}_pmc_summary.txt")).read())
