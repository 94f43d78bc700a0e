#!/usr/bin/env python3
"""MG-GCR solve on a chosen lattice (bench.py's run_mg), for profiling: tools/mg_profile.py 16,16,16,16"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    X = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,16,16,16").split(","))
    qa = importlib.import_module("quda-qkxtm-multigrid_amd")
    qa.init(0)
    print(json.dumps(bench.run_mg(qa, X)))
    qa.end()


if __name__ == "__main__":
    main()
