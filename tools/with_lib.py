#!/usr/bin/env python
"""Run a script against another build of the library (A/B on one box):  MPOSE_LIB=path python tools/with_lib.py bench.py --steps 20 ..."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import margipose_amd.build as b
if os.environ.get('MPOSE_LIB'):
    b.LIB_PATH = os.path.abspath(os.environ['MPOSE_LIB'])
    b.is_stale = lambda: False
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
