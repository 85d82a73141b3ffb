#!/usr/bin/env python3
"""One conv shape a few times (for rocprofv3 --pmc runs): python tools/one_conv.py B H Ci Co mode iters"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import conv_bench  # noqa: E402
B, H, Ci, Co, mode, iters = (int(v) for v in sys.argv[1:7])
conv_bench.run(B, H, Ci, Co, mode, iters)
