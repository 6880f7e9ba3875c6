#!/usr/bin/env python3
"""Register / scratch use of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: tools/kres.py marl_amd/csrc/agent.hip [name filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-c", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
if not rows:                      # compile error: c++filt without arguments would wait on stdin
    sys.exit("no kernels reported - compile output:\n" + err[-3000:])
names = subprocess.run(["/usr/bin/c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
print("%-90s %5s %5s %6s %6s %8s %4s" % ("kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "occ"))
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*\)$", "", n)
    if filt and filt not in n:
        continue
    print("%-90s %5s %5s %6s %6s %8s %4s" % (n[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy")))
