"""bench.py against the tuning build of the library (A/B experiments; the shipped library is the default everywhere else):
    python tools/bench_with_lib.py [bench.py arguments]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopticalflow_amd import _lib, build   # noqa: E402

_lib.LIB_PATH = os.environ.get('UNFLOW_LIB_PATH') or build.LIB_TUNING      # (UNFLOW_LIB_PATH: a tagged variant build)
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
