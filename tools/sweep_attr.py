"""bench.py with engine / ops module attributes overridden: python tools/sweep_attr.py engine.BWD_CHUNKS=8 ops.X=1 -- <bench args>
(development tool: the tuned constants are plain attributes, not switches)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sep = sys.argv.index('--') if '--' in sys.argv else len(sys.argv)
sets, rest = sys.argv[1:sep], sys.argv[sep + 1:]
for s in sets:
    name, val = s.split('=')
    mod, attr = name.rsplit('.', 1)
    m = importlib.import_module('gantron_amd.' + mod)
    assert hasattr(m, attr), name
    setattr(m, attr, type(getattr(m, attr))(eval(val)))
import bench
sys.argv = ['bench.py'] + rest
bench.main()
