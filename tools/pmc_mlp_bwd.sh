#!/bin/bash
# Counter passes over tools/pmc_mlp_bwd.py (GPU box): the backward GEMM families alone (SQ cycles / instruction mix / LDS / HBM bytes,
# each in its own rocprofv3 --pmc pass) and their durations alone vs beside the other stream (--kernel-trace).
# Output: gpurun_out/pmc_bwd/{a,b,c,d,e}.txt + alone_stats.txt / beside_stats.txt; tools/pmc_mlp_bwd_summary.py composes profiles/rNN_pmc_mlp_bwd.txt.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bwd; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $O/$n -o p -- python3 $R/tools/pmc_mlp_bwd.py alone > $O/$n.log 2>&1
  db=$(find $O/$n -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python3 $R/tools/rocpd_pmc.py $db > $O/$n.txt 2>&1; fi
  rm -rf $O/$n
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU
run c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run d FETCH_SIZE
run e WRITE_SIZE
for m in alone beside; do
  timeout 300 rocprofv3 --kernel-trace -d $O/$m -o p -- python3 $R/tools/pmc_mlp_bwd.py $m > $O/$m.log 2>&1
  db=$(find $O/$m -name "*.db" 2>/dev/null | head -1)
  if [ -n "$db" ]; then python3 $R/tools/rocpd_stats.py $db 30 > $O/${m}_stats.txt 2>&1; fi
  rm -rf $O/$m
done
cd $R
python3 tools/pmc_mlp_bwd.py time > $O/time.txt 2>&1
tail -20 $O/time.txt
grep -h "mlp_\|gram\|^kernel" $O/a.txt | cut -c1-250
