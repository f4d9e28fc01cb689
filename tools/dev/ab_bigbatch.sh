#!/bin/bash
# On the GPU box: the large-batch train step (65,536 pairs, 960 MB table) timed with the in-tree library; usage: tools/dev/ab_bigbatch.sh TAG
TAG=${1:-run}
timeout -k 10 400 python tools/bigbatch_probe.py 1200000 65536 > gpurun_out/bb_$TAG.log 2>&1 && timeout -k 10 400 python tools/bigbatch_probe.py 1200000 65536 >> gpurun_out/bb_$TAG.log 2>&1
cat gpurun_out/bb_$TAG.log
