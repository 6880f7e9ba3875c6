#!/usr/bin/env python3
"""Thread sweep of the CPU baseline (the oracle's learner update + batched rollout, bench.cpu_baseline) on this box's
host cores: profiles/<tag>_cpu_threads.txt.  BASELINE.md section 4 asks for torch threads = os.cpu_count(); the oracle's
ops are small, so more threads are not always faster - bench.py reports the fastest of 16 / 64 / all."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    host = os.cpu_count() or 1
    if len(sys.argv) <= 2:
        print("host CPUs %d (%s); QMIX 2s3z, %d envs x T=120, oracle train() + batched rollout" % (host, bench.cpu_model(), envs), flush=True)
    import subprocess
    for c in sorted({min(host, x) for x in (8, 16, 32, 64, 128, host)}):
        if len(sys.argv) > 2:      # child: one thread count
            break
        # each thread count in a child process under a time limit (an oversubscribed torch thread pool can take minutes)
        try:
            out = subprocess.run([sys.executable, __file__, str(envs), str(c)], capture_output=True, text=True, timeout=150).stdout
            print(out.strip().splitlines()[-1] if out.strip() else "threads %3d : no output" % c, flush=True)
        except subprocess.TimeoutExpired:
            print("threads %3d : did not finish 150 s (oversubscribed thread pool)" % c, flush=True)
    if len(sys.argv) > 2:
        c = int(sys.argv[2])
        r = bench.cpu_baseline("qmix", "2s3z", 120, envs, budget_s=8, threads=c)
        print("threads %3d : learner %.2f updates/s (%.0f transitions/s)  batched rollout %.0f env-steps/s  pipeline %.0f env-steps/s"
              % (c, r["learner_updates_per_sec"], r["learner_transitions_per_sec"], r["batched_rollout_env_steps_per_sec"], r["value"]))
