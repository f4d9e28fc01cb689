#!/usr/bin/env python3
"""Runs bench.py's score_kernel_roofline() alone (fused gather + ComplEx score, d=200, 960 MB table,
4 M triples per launch) so that rocprofv3 can attribute kernel time and HBM counters to it."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.score_kernel_roofline(200, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)))
