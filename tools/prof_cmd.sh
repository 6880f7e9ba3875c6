#!/bin/bash
# kernel-time table of any python tool ON THE GPU BOX: bash tools/prof_cmd.sh <filter regex> tools/<script>.py [args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=$1; shift
rm -rf /tmp/pc
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o p -- python3 "$@" > /tmp/pc.log 2>&1
python3 - "$F" <<'PY'
import csv, glob, re, sys
f = glob.glob('/tmp/pc/**/*kernel_stats.csv', recursive=True)
if not f:
    sys.exit("no stats: " + open('/tmp/pc.log').read()[-2000:])
for r in csv.DictReader(open(f[0])):
    if re.search(sys.argv[1], r['Name']):
        n = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        print("%-60s calls %5s avg %8.1f us" % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
