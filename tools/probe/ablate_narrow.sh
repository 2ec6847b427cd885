#!/bin/bash
# sa1's narrow GEMMs (tools/probe/bench_narrow.py) on ablation builds of mlp_fast.hip (-DBF3_ABL=n: 1 no MFMAs, 2 no epilogue, 8 no staging;
# sums combine; results wrong by construction, only the time is read).  Libraries: tools/probe/lib/libvotenet_abl<n>.so
timeout 200 python tools/probe/bench_narrow.py 2>&1 | grep "variant\|narrow\|plain"
for a in 1 2 3 8 10 11; do VARIANT=abl$a timeout 200 python tools/probe/bench_narrow.py 2>&1 | grep "variant\|narrow\|plain"; done
