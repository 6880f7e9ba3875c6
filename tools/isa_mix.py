#!/usr/bin/env python3
"""Instruction mix of the kernels of one .hip file (hipcc -save-temps ISA): per kernel the static counts of MFMA, other vector ALU,
LDS, vector memory, scratch (spill) and waitcnt instructions.  usage: tools/isa_mix.py file.hip [name filter]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); filt = sys.argv[2] if len(sys.argv) > 2 else ""
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-c", "-save-temps", "-o", os.path.join(d, "o.o"), src], cwd=d, capture_output=True)
asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
cur, stats = None, {}
for line in open(os.path.join(d, asm)):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); stats[cur] = dict(mfma=0, valu=0, lds=0, vmem=0, scratch=0, wait=0, salu=0, barrier=0); continue
    if cur is None: continue
    t = line.strip().split()
    if not t or t[0].startswith((".", ";")) or t[0].endswith(":"): continue
    op = t[0]
    s = stats[cur]
    if op.startswith("v_mfma"): s["mfma"] += 1
    elif op.startswith("scratch_"): s["scratch"] += 1
    elif op.startswith("ds_"): s["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): s["vmem"] += 1
    elif op.startswith("v_"): s["valu"] += 1
    elif op == "s_waitcnt": s["wait"] += 1
    elif op == "s_barrier": s["barrier"] += 1
    elif op.startswith("s_"): s["salu"] += 1
names = subprocess.run(["/usr/bin/c++filt"] + list(stats), capture_output=True, text=True).stdout.splitlines()
print("%-70s %6s %6s %6s %6s %7s %6s %6s %7s" % ("kernel", "mfma", "valu", "lds", "vmem", "scratch", "wait", "salu", "barrier"))
for (k, s), n in zip(stats.items(), names):
    n = re.sub(r"\(.*\)$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
    if filt and filt not in n: continue
    print("%-70s %6d %6d %6d %6d %7d %6d %6d %7d" % (n[:70], s["mfma"], s["valu"], s["lds"], s["vmem"], s["scratch"], s["wait"], s["salu"], s["barrier"]))
