#!/bin/bash
# same-box alternating A/B: old library vs new (split-K off) vs new (split-K on)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  VARIANT=old timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  TOGGLES="mlp.SPLIT_K=False" timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
done
