#!/bin/bash
# Same-box A/B against an earlier commit: tools/ab_against_commit.sh <commit>
# unpacks that commit into _abtree/<commit>/ (git-ignored, but shipped to the GPU box by gpurun), builds its library there, and prints the
# gpurun line that alternates its bench.py with the current tree's in ONE call.  Box-to-box spread of the same binary is +-3 % per
# step: a comparison across calls cannot see a 1-2 % change (round 3: the pre-scaled gate math cost the headline 1 % and went
# unnoticed for most of the round - profiles/archive/r03_prescale_ab.txt).  Remove _abtree/ afterwards.
set -e
C=${1:?commit}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
D="$ROOT/_abtree/$C"
rm -rf "$D"; mkdir -p "$D"
git -C "$ROOT" archive "$C" | tar -x -C "$D"
rm -rf "$D/profiles" "$D/tests/golden"
make -j8 -C "$D/marl_amd/csrc" > /dev/null
echo "built $D/marl_amd/libmarl_hip.so"
cat <<EOT
gpurun --timeout 900 -- 'for i in 1 2 3; do echo -n "now: "; python bench.py --no-cpu-baseline --no-configs 2>/dev/null | grep "^{\"metric\"" | cut -c1-120;
  echo -n "$C: "; (cd _abtree/$C && python bench.py --no-cpu-baseline 2>/dev/null | grep "^{\"metric\"" | cut -c1-120); done'
EOT
