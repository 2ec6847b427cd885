#!/bin/bash
# Counter passes over tools/pmc_pieces.py (GPU box): every kernel of one SA level's forward + backward on the piece layout, alone on one
# stream: durations (--kernel-trace), HBM bytes (FETCH_SIZE | WRITE_SIZE, separate passes), MFMA busy cycles.
# Output: gpurun_out/pmc_pieces/<level>_{trace,fetch,write,sq}.txt; tools/pmc_pieces_summary.py composes profiles/rNN_pmc_pieces.txt.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_pieces; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lv in sa2 sa1; do
  timeout 300 rocprofv3 --kernel-trace -d $O/t -o p -- python3 $R/tools/pmc_pieces.py $lv > $O/${lv}_trace.log 2>&1
  db=$(find $O/t -name "*.db" | head -1); python3 $R/tools/rocpd_stats.py $db 60 > $O/${lv}_trace.txt 2>&1; rm -rf $O/t
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
    set -- $pass; n=$1; shift
    timeout 300 rocprofv3 --pmc "$@" -d $O/p -o p -- python3 $R/tools/pmc_pieces.py $lv > $O/${lv}_$n.log 2>&1
    db=$(find $O/p -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $db > $O/${lv}_$n.txt 2>&1; rm -rf $O/p
  done
  grep "balls" $O/${lv}_trace.log
done
