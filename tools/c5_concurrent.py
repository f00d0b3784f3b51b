"""BASELINE C5 as worded, stand-alone: bench.py's two-concurrent-instances leg without the rest of the line.
usage: python3 tools/c5_concurrent.py [seconds] [bins]"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
print(json.dumps(b.c5_concurrent_figures(float(sys.argv[1]) if len(sys.argv) > 1 else 2.0, int(sys.argv[2]) if len(sys.argv) > 2 else 512), indent=1))
