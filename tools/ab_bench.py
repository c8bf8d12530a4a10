#!/usr/bin/env python3
"""Development aid: bench.py against another build of the library (A/B of compile-time switches).
usage: python tools/ab_bench.py <path/to/libvariant.so> [bench.py arguments]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
