#!/bin/bash
# round 3: row-streamed cost-volume backward against the group-split ring kernel (tuning library)
out=gpurun_out/r3
mkdir -p $out
UNFLOW_RS_VARIANTS=${UNFLOW_RS_VARIANTS:-7,9} UNFLOW_MICROBENCH_TUNING=1 timeout 120 python3 tools/microbench.py corr_bwd_rs > $out/corr_bwd_rs.txt 2>&1
cat $out/corr_bwd_rs.txt
