cd $GRAFT_REPO_ROOT
for cfg in "2 192" "1 192" "1 128" "1 256"; do set -- $cfg
  HOOKS="votenet_debug_sparse_centre_teams=$1 votenet_debug_sparse_centre_workgroups=$2" bash tools/serial_last_step.sh; echo "teams $1 cap $2: $(grep sparse gpurun_out/serial/last_step.txt | awk '{printf "%s ", $(NF-1)}')"
done
for rep in 1 2 3; do
  for cfg in "2 192" "1 192" "1 128"; do set -- $cfg
    HOOKS="votenet_debug_sparse_centre_teams=$1 votenet_debug_sparse_centre_workgroups=$2" timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  done
done
