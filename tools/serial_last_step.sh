R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/serial; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/t -o s -- python3 $R/tools/serial_step.py > $O/traced.txt 2>&1
db=$(find $O/t -name "*.db" | head -1)
python3 $R/tools/rocpd_last_step.py $db > $O/last_step.txt 2>&1
rm -rf $O/t
