#!/usr/bin/env python3
"""GPU check: rounds of the CF session import (bench.cf_write_path) with SMATRIX_TRACE_ROUNDS=1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
r = bench.cf_write_path(torch, torch.device("cuda", 0))
print(r)
