#!/bin/bash
# three concurrent det_step.py processes (the GPU shared between them); prints each one's last line.  usage: det3.sh [steps]
n=${1:-40}
python tools/probe/det_step.py $n > /tmp/d1.txt 2>&1 &
p1=$!
python tools/probe/det_step.py $n > /tmp/d2.txt 2>&1 &
p2=$!
python tools/probe/det_step.py $n > /tmp/d3.txt 2>&1
wait $p1 $p2
for f in /tmp/d1.txt /tmp/d2.txt /tmp/d3.txt; do tail -n 1 $f; done
