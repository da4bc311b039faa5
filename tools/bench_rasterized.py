"""Thin wrapper kept for the profiling scripts: the rasterised streaming benchmark lives in bench.py --mode rasterized."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raise SystemExit(subprocess.call([sys.executable, os.path.join(REPO, "bench.py"), "--mode", "rasterized", *sys.argv[1:]]))
