#!/bin/bash
# two --kernel-trace runs of tools/probe/trace_step.py with different TOGGLES; per-queue timelines of one steady step of each
# usage: bash tools/trace_ab.sh NAME "TOGGLES_A" "TOGGLES_B"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_ab_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for tg in "$2" "$3"; do
  i=$((i+1))
  TOGGLES="$tg" timeout 300 rocprofv3 --kernel-trace -d $O/t$i -o s -- python3 $R/tools/probe/trace_step.py > $O/run$i.log 2>&1
  db=$(find $O/t$i -name "*.db" | head -1)
  python3 $R/tools/rocpd_step.py $db 8 > $O/step$i.txt 2>&1
  python3 $R/tools/rocpd_stats.py $db 60 > $O/stats$i.txt 2>&1
  rm -rf $O/t$i
  echo "== $tg"; tail -1 $O/run$i.log; head -3 $O/step$i.txt
done
