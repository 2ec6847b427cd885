#!/bin/bash
# SQ counter passes over tools/probe/bf3_one.py (GPU box): where do the waves of the BF3 GEMM spend their cycles?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_bf3; mkdir -p $O
ARGS="${@:-524288 128 256 64}"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/a -o p -- python3 $R/tools/probe/bf3_one.py $ARGS > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU -d $O/b -o p -- python3 $R/tools/probe/bf3_one.py $ARGS > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/c -o p -- python3 $R/tools/probe/bf3_one.py $ARGS > $O/c.log 2>&1
cd $R
for d in a b c; do db=$(ls $O/$d/*.db 2>/dev/null | head -1); if [ -n "$db" ]; then python tools/rocpd_pmc.py $db > $O/$d.txt 2>&1; fi; done
rm -rf $O/*/*.db
grep -h "mlp_linear_fast\|^kernel" $O/a.txt $O/b.txt $O/c.txt | cut -c1-300
