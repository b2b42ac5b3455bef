"""Run a tools/ script or bench.py against an alternative build of the library: QT_ALT_LIB=tools/micro/<name>.so (diagnostics)."""
import os, runpy, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'quadtree-mpnnlstm_amd'))
from qtmpnn import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ['QT_ALT_LIB'])
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
