"""Round-6 working script: the headline BA line (configs[4], 20 LM iterations, no CPU legs) under a list of environment variants.
usage: python tools/r06_ba_variants.py "OSFM_BA_FORK=2" "OSFM_BA_CS=10" ...   (the empty string = the default)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, json
sys.path.insert(0, %r)
import bench_ba
from opensfm_amd import _lib
r = bench_ba.run(_lib.default_context(0), cpu_baseline=False, grid=False)
print(json.dumps({"value": r["value"], "ms": r["lm_iteration"]["ms"], "pcg": r["pcg_iterations"], "matvec_ms": r["roofline"]["avg_matvec_ms"]}))
""" % ROOT
out = {}
for variant in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in variant.split():
        k, v = kv.split("=", 1)
        env[k] = v
    rows = []
    for rep in range(2):
        p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        rows.append(json.loads(line[-1]) if line else {"error": p.stderr[-300:]})
    out[variant or "default"] = rows
print(json.dumps(out, indent=1))
