#!/bin/bash
# same-box alternating A/B: the arg-max gather of the pooled layers' weight gradient walking pieces vs centres (mlp.POOL_WGRAD_CENTRES)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  TOGGLES='mlp.POOL_WGRAD_CENTRES=False' timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  for w in 128 192 256 384; do
    HOOKS="votenet_debug_sparse_centre_workgroups=$w" timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  done
done
