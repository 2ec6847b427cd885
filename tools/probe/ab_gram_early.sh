#!/bin/bash
# same-box alternating A/B of model.GRAM_EARLY (tools/probe/variant_step.py)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  TOGGLES='model.GRAM_EARLY=("sa1",)' timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  TOGGLES='model.GRAM_EARLY=("sa1","sa2")' timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  TOGGLES='model.GRAM_EARLY=("sa2","sa1")' timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
done
