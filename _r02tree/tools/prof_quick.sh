#!/bin/bash
# quick kernel-time table of one learner configuration ON THE GPU BOX: bash tools/prof_quick.sh [prof_learner args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o p -- python3 tools/prof_learner.py --warmup 3 --updates 20 "$@" > /tmp/pq.log 2>&1
tail -1 /tmp/pq.log
python3 tools/kstats.py "$(find /tmp/pq -name '*kernel_stats.csv' | head -1)" 20 ${TOPN:-14}
