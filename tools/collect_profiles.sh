#!/bin/bash
# Run on the GPU box (gpurun): kernel traces of the train / forward bench and the PMC passes behind bench.py's
# roofline objects.  Everything lands under gpurun_out/; the summaries are then copied into profiles/ by hand.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/train -o t -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/train.log 2>&1
rocprofv3 --kernel-trace -d $O/fwd -o t -- python3 $R/bench.py --workload fwd --steps 10 --warmup 2 --no-cpu-baseline > $O/fwd.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fps_fetch -o p -- python3 $R/tools/pmc_fps.py > $O/fps_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/fps_write -o p -- python3 $R/tools/pmc_fps.py > $O/fps_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/mlp_sq -o p -- python3 $R/tools/pmc_mlp.py > $O/mlp_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/mlp_fetch -o p -- python3 $R/tools/pmc_mlp.py > $O/mlp_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/mlp_write -o p -- python3 $R/tools/pmc_mlp.py > $O/mlp_write.log 2>&1
cd $R
for d in train fwd; do python tools/rocpd_stats.py $(ls $O/$d/*.db | head -1) 60 > $O/${d}_kernel_stats.txt; done
for d in fps_fetch fps_write mlp_sq mlp_fetch mlp_write; do python tools/rocpd_pmc.py $(ls $O/$d/*.db | head -1) > $O/${d}.txt 2>&1; done
python bench.py > $O/train_bench_line.json 2> $O/train_bench.err
python bench.py --workload fwd > $O/fwd_bench_line.json 2> $O/fwd_bench.err
rm -rf $O/*/*.db   # keep the merge small: the summaries are what is judged
tail -c 600 $O/train_bench_line.json
