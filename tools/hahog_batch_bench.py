"""HAHOG leg of bench.py alone: single-image call, batches (host and device-resident images), the compiled reference beside it."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from opensfm_amd import _lib

ctx = _lib.Context()
print(json.dumps(bench.hahog_bench(ctx, "--cpu" in sys.argv)))
