#!/usr/bin/env python3
"""python tools/summarize_profiles.py <round>: condenses gpurun_out/<round>_* (tools/refresh_profiles.sh <round> on the GPU box) into
committed evidence under profiles/: <round>_bench_kernel_stats.csv, <round>_bench_line.json, <round>_bench_full.json,
<round>_pmc_<workload>.json (one per bench workload, with the library version they were taken on), kernel stats of the
QPLEX / QTRAN / MMM2 updates, timings, soak, shard steps."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    sys.exit(__doc__)
G, P, tag = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles"), sys.argv[1]


def cp(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, dst)
        return True
    return False


cp(os.path.join(G, tag + "_bench", "p_kernel_stats.csv"), os.path.join(P, tag + "_bench_kernel_stats.csv"))
for sub in ("qplex_f32", "qplex_bf16x6", "qmix_bf16x6", "qtran", "mmm2_bf16"):
    cp(os.path.join(G, "%s_%s" % (tag, sub), "p_kernel_stats.csv"), os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, sub)))
for name in ("_bench_line.json", "_bench_full.json", "_bench_profiled_line.json"):
    src = os.path.join(G, tag + name)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        line = [l for l in open(src).read().splitlines() if l.startswith("{")][-1].strip()      # (stdout also carries the host classes' prints)
        json.loads(line)
        open(os.path.join(P, tag + name), "w").write(line + "\n")
for name in ("_mlp3_times.txt", "_rollout_times.txt", "_stamps.txt", "_reducer_world1.txt", "_unroll_x6_times.txt", "_bptt_x6_ab.txt", "_learner_rates.txt", "_shard_steps.txt", "_soak.txt", "_qmix_times.txt"):
    cp(os.path.join(G, tag + name), os.path.join(P, tag + name))
vf = os.path.join(G, tag + "_lib_version.txt")
ver = open(vf).read().strip() if os.path.exists(vf) else None


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


for d in sorted(glob.glob(os.path.join(G, tag + "_pmc_*"))):
    if not os.path.isdir(d):
        continue
    w = os.path.basename(d)[len(tag) + 5:]
    per, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "pass*", "p_counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k.startswith(("at::", "__amd", "elementwise", "vectorized")) or "at::native" in r["Kernel_Name"]:
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"], f)
            if key not in seen:
                seen.add(key)
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not per:
        print("no counters under", d)
        continue
    out = {"workload": w, "lib_version": ver,
           "source": "rocprofv3 --pmc <counter group> --kernel-trace, separate passes (FETCH_SIZE | WRITE_SIZE | SQ group) of tools/prof_learner.py "
                     "at this workload (tools/refresh_profiles.sh, 1x MI355X); values are means per launch",
           "hbm_correction": "gfx950: FETCH_SIZE counts half of wide (16 B/lane) coalesced reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB "
                             "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "kernels": {}}
    for k, cs in sorted(per.items()):
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
        if dur[k]:
            e["avg_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(P, "%s_pmc_%s.json" % (tag, w)), "w"), indent=1, sort_keys=True)
    top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0) * kv[1]["launches"])[:4]
    print("%-44s %d kernels; top HBM: %s" % (w, len(out["kernels"]), ", ".join("%s %.3g B" % (k[:36], v.get("hbm_bytes_per_launch", float("nan"))) for k, v in top)))
if os.path.exists(os.path.join(G, "parity_margins.txt")):
    pass      # (copied by hand: the fp32 and the bf16x6 runs of the suite write the same file)
