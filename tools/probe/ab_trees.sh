#!/usr/bin/env bash
# probe: the pipelined train step of THIS tree against an older tree unpacked under tools/probe/old_tree (git archive <commit> votenet_amd
# include tools/probe/variant_step.py | tar -x -C tools/probe/old_tree; not committed), alternating processes on one box.
set -u
cd "$(dirname "$0")/../.."
(cd tools/probe/old_tree && bash votenet_amd/csrc/build.sh > /dev/null 2>&1) || { echo "old tree build failed"; exit 1; }
for rep in $(seq 1 ${REPS:-4}); do
  echo "old: $(cd tools/probe/old_tree && python tools/probe/variant_step.py 2>&1 | tail -1)"
  echo "new: $(python tools/probe/variant_step.py 2>&1 | tail -1)"
done
