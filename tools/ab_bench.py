import os, sys
sys.path.insert(0, os.getcwd())
from mc3_amd import _lib
_lib.LIB_PATH = os.environ['AB_LIB']
sys.argv = ['bench.py', '--no-cpu-baseline']
import runpy
runpy.run_path('bench.py', run_name='__main__')
