#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>_* (written by tools/refresh_profiles.sh on the GPU box) into the committed
evidence under profiles/:  <tag>_bench_kernel_stats.csv, <tag>_bench_line.json, <tag>_pmc.json, <tag>_stamps.txt."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")

shutil.copy(os.path.join(G, tag + "_bench", "p_kernel_stats.csv"), os.path.join(P, tag + "_bench_kernel_stats.csv"))
for name in ("_bench_line.json", "_bench_full_line.json"):
    src = os.path.join(G, tag + name)
    if os.path.exists(src):
        line = open(src).read().strip()
        json.loads(line)
        open(os.path.join(P, tag + name), "w").write(line + "\n")
shutil.copy(os.path.join(G, tag + "_stamps.txt"), os.path.join(P, tag + "_stamps.txt"))
# QPLEX (BASELINE config 3 shape at 4096 envs): kernel stats of tools/prof_learner.py --alg qplex + the standalone
# timings of the fused head kernels (tools/time_mlp3.py)
if os.path.exists(os.path.join(G, tag + "_qplex", "p_kernel_stats.csv")):
    shutil.copy(os.path.join(G, tag + "_qplex", "p_kernel_stats.csv"), os.path.join(P, tag + "_qplex_kernel_stats.csv"))
if os.path.exists(os.path.join(G, tag + "_mlp3_times.txt")):
    shutil.copy(os.path.join(G, tag + "_mlp3_times.txt"), os.path.join(P, tag + "_mlp3_times.txt"))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def pmc_summary(sub, names, source, dest):
    """means per launch of the PMC passes under gpurun_out/<tag><sub>/pass*/ for kernels whose name contains one of `names`"""
    per = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> per-dispatch values
    dur = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(G, tag + sub, "pass*", "p_counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not any(x in k for x in names):
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            d = (r["Dispatch_Id"], f)
            if d not in seen:
                seen.add(d)
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if not per:
        return None
    out = {"source": source,
           "hbm_correction": "gfx950: FETCH_SIZE counts half of wide (16 B/lane) coalesced reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB "
                             "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
           "kernels": {}}
    for k, cs in sorted(per.items()):
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs (256 CUs x 4)
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
        if dur[k]:
            e["avg_ns_under_pmc"] = sum(dur[k]) / len(dur[k])
            if "GRBM_GUI_ACTIVE" in e:
                e["clock_ghz_est"] = e["GRBM_GUI_ACTIVE"] / 8.0 / e["avg_ns_under_pmc"]
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(P, tag + dest), "w"), indent=1, sort_keys=True)
    for k, v in out["kernels"].items():
        print("%-44s launches %3d  mfma_busy %.3f  clock %.2f GHz  hbm %.3g B" % (k[:44], v["launches"], v.get("mfma_busy_frac", float("nan")),
              v.get("clock_ghz_est", float("nan")), v.get("hbm_bytes_per_launch", float("nan"))))
    return out


out = pmc_summary("_pmc", ("agent_", "qmix_fused_kernel", "wgrad_direct", "wgrad_tall", "synth_rollout"),
                  "rocprofv3 --pmc <group> --kernel-trace, four separate passes of tools/prof_learner.py --updates 3 --rollouts 2 "
                  "(QMIX 2s3z, 4096 envs, T=120, 1x MI355X); values are means per launch", "_pmc.json")
if out:
    fw = [v for k, v in out["kernels"].items() if k.startswith("agent_fwd_kernel")]
    if fw:
        tot = sum(v.get("hbm_bytes_per_launch", 0) * v["launches"] for v in fw)
        n = sum(v["launches"] for v in fw)
        print("agent_fwd avg HBM bytes per launch: %.4g over %d launches" % (tot / max(n, 1), n))
pmc_summary("_pmcq", ("mlp3_", "qplex_mix"),
            "rocprofv3 --pmc <group> --kernel-trace, four separate passes of tools/prof_learner.py --alg qplex --updates 3 "
            "(QPLEX 2s3z, 4096 envs, T=120, 1x MI355X; tools/pmc_qplex.sh); values are means per launch", "_pmc_qplex.json")

# ---- round-2 additions (tools/refresh_profiles_r02.sh)
for sub, dst in (("_qtran", "_qtran_kernel_stats.csv"), ("_mmm2_fp32", "_mmm2_fp32_kernel_stats.csv"), ("_mmm2_bf16", "_mmm2_bf16_kernel_stats.csv")):
    f = os.path.join(G, tag + sub, "p_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, tag + dst))
for name in ("_bench_mmm2_fp32_line.json", "_bench_mmm2_bf16_line.json", "_learner_rates.txt", "_shard_steps.txt",
             "_bw_probe.txt", "_coissue_probe.txt", "_mfma_peak_probe.txt", "_bf16x3_probe.txt", "_gaps.txt"):
    src = os.path.join(G, tag + name)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, tag + name))
pmc_summary("_pmcw", ("qmix_wide",),
            "rocprofv3 --pmc <group> --kernel-trace, separate passes of tools/prof_learner.py --alg qmix --shape MMM2 --envs 1024 "
            "--mixer-dtype bf16 (1x MI355X); values are means per launch", "_pmc_qmix_wide.json")
if os.path.exists(os.path.join(G, "parity_margins.txt")):
    shutil.copy(os.path.join(G, "parity_margins.txt"), os.path.join(P, tag + "_parity_margins.txt"))
