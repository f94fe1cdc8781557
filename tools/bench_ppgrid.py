"""GPU box: bench.py with the pair kernels confined to N work-groups (does leaving CUs to the other lane's launches pay?).
usage: python tools/bench_ppgrid.py N [bench.py flags]"""
import sys
sys.path.insert(0, ".")
import bench                                   # (its environment set-up first)
from phiseg_code_amd import engine, runtime as rt
n = int(sys.argv[1])
_init = engine.Plan.__init__
def init(self, *a, **k):                       # the first plan is built after bench.main() has brought the device up
    rt.lib().debug_pair_kernel_grid(n)
    _init(self, *a, **k)
engine.Plan.__init__ = init
sys.argv = ["bench.py"] + sys.argv[2:]
bench.main()
