"""Which parameter gradients of the FIRST executor step of a fresh process differ from the operator path's?  (first-use
flake hunt of the backward side stream; run many times: `for i in ...; do python scripts/side_bwd_diag.py; done`)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_trunk import _batch, _model, _grads
from lidog_amd import me as ME, trunk
from lidog_amd.trainer import LiDOGStep
from lidog_amd.optim import make_optimizer

ME.set_backward_overlap(True)
trunk.set_fusions(7)
runs = {}
for on in (False, True):
    trunk.set_enabled(on)
    model = _model()
    step = LiDOGStep(model, make_optimizer("Adam", model, 1e-2, weight_decay=1e-4))
    out = step.training_step(_batch((61, 71)))
    torch.cuda.synchronize()
    runs[on] = (float(out["loss"]), _grads(model))
bad = []
for k, g in runs[True][1].items():
    r = runs[False][1][k]
    if g is None or r is None:
        continue
    if not torch.equal(g, r):
        d = (g - r).abs()
        bad.append("%s shape %s: %d of %d differ, max %.3e, executor zeros %d, ref absmax %.3e" % (
            k, tuple(g.shape), int((d > 0).sum()), g.numel(), d.max().item(), int((g == 0).sum()), r.abs().max().item()))
print("DIAG loss", runs[True][0] == runs[False][0], "bad", len(bad))
for b in bad:
    print("DIAG  ", b)
