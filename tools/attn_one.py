#!/usr/bin/env python3
"""One attention shape, a few launches (for rocprofv3 --pmc):  python tools/attn_one.py B H T hd"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaw_amd  # noqa
import importlib.util
spec = importlib.util.spec_from_file_location("attn_bench", os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_bench.py"))
ab = importlib.util.module_from_spec(spec); spec.loader.exec_module(ab)
B, H, T, hd = (int(x) for x in sys.argv[1:5])
ab.run("shape", B, H, T, hd, 3)
