"""Dev tool: what keeping k CUs out of the persistent GEMM grids (uvtg_set_reserved_cus, what TrainStep(comm_cus=k) does on multi-rank jobs)
costs a single-GPU step: bench.py's step with the reservation forced after TrainStep is built.  usage: reserved_cus_probe.py k [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
k = int(sys.argv[1]); sys.argv = [sys.argv[0]] + sys.argv[2:]
from univtg_amd import trainer
_init = trainer.TrainStep.__init__
def init(self, *a, **kw):
    _init(self, *a, **kw)
    self.gemm_cus = self.lib.uvtg_set_reserved_cus(k)
    print(f"[reserved_cus_probe] {k} CUs reserved -> GEMM grids sized for {self.gemm_cus} CUs", file=sys.stderr)
trainer.TrainStep.__init__ = init
import bench
bench.main()
