#!/bin/bash
# the GPU test suite in three concurrent processes on one GPU (kernels of different processes -- MFMA GEMMs among them -- share compute
# units): sporadic hardware-level hazards show up as failures of the bit-exact tests.  usage: suite3.sh [pytest args]
python -m pytest tests -m gpu -q -x -p no:cacheprovider "$@" > /tmp/s1.txt 2>&1 &
p1=$!
python -m pytest tests -m gpu -q -x -p no:cacheprovider "$@" > /tmp/s2.txt 2>&1 &
p2=$!
python -m pytest tests -m gpu -q -x -p no:cacheprovider "$@" > /tmp/s3.txt 2>&1
wait $p1 $p2
for f in /tmp/s1.txt /tmp/s2.txt /tmp/s3.txt; do tail -n 3 $f; done
