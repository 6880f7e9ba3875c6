#!/usr/bin/env python3
"""Compresses a kernel's ISA into its sequence of memory / wait / MFMA events:  L<n> global loads, S<n> global stores,
D<n> LDS ops (optional), W<c> s_waitcnt vmcnt(c), M<n> MFMAs, BAR barriers.  Makes "a load followed at once by a wait for
it" (masking or selecting on a loaded value inside the prefetch code) and vmcnt(0) drains visible at a glance.

    python tools/isa_events.py marl_amd/csrc/gemm.hip wgrad_kernel [-D...]
"""
import subprocess
import sys


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    out = "/tmp/isa_events.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-S", "-o", out, src, "--cuda-device-only"] + extra,
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
    names = [l.split(":")[0] for l in s.split("\n") if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
    for name in names:
        i = s.index(name + ":")
        j = s.index(".Lfunc_end", i)
        k = [l.strip() for l in s[i:j].split("\n")]
        k = [l for l in k if l and not l.startswith(";") and not l.startswith(".")]
        ev = []
        for l in k:
            if "mfma" in l: ev.append("M")
            elif l.startswith("global_load") or l.startswith("buffer_load"): ev.append("L")
            elif l.startswith("global_store") or l.startswith("buffer_store"): ev.append("S")
            elif "vmcnt" in l: ev.append("W" + l.split("vmcnt(")[1].split(")")[0] + "_")
            elif l.startswith("s_barrier"): ev.append("BAR")
        seq, last, cnt = [], None, 0
        for t in ev:
            if t == last: cnt += 1
            else:
                if last: seq.append("%s%d" % (last, cnt) if not last.startswith("W") else last + ("x%d" % cnt if cnt > 1 else ""))
                last, cnt = t, 1
        if last: seq.append("%s%d" % (last, cnt) if not last.startswith("W") else last)
        print("== %s (%d instructions)" % (name, len(k)))
        print(" ".join(seq))


if __name__ == "__main__":
    main()
