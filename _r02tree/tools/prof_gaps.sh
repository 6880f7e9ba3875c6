#!/bin/bash
# device idle gaps of the default bench step ON THE GPU BOX: bash tools/prof_gaps.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pg
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o p -- python3 bench.py --no-cpu-baseline --steps 12 --warmup 4 "$@" > /tmp/pg.log 2>&1
grep '^{"metric"' /tmp/pg.log | tail -1 | cut -c 1-160
python3 tools/gaps.py "$(find /tmp/pg -name '*kernel_trace.csv' | head -1)" ${NLAST:-600}
