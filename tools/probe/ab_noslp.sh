#!/bin/bash
# same-box alternating A/B: the library as built vs the GEMM sources (or all hot sources) compiled with -fno-slp-vectorize
for rep in 1 2 3; do
  timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  VARIANT=noslp_gemm timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
  VARIANT=noslp_all timeout 300 python tools/probe/variant_step.py 2>&1 | grep "^variant"
done
