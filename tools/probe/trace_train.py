"""Run tools/train.py with periodic all-thread stack dumps (where does a slow iteration spend its time?)."""
import faulthandler
import os
import runpy
import sys

faulthandler.dump_traceback_later(int(os.environ.get('DUMP_EVERY', '45')), repeat=True, file=sys.stderr)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [os.path.join(root, 'tools', 'train.py')] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
