"""Layer-by-layer gradient comparison GPU vs CPU oracle (same wiring), reverse order."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import GOLDEN, seeded_state_dict, small_batch
import lidog_amd, lidog_amd.me as ME
from lidog_amd.losses import SoftDICELoss, DICELoss
import oracle.me_cpu as OME
from oracle.ref_torch import Encoder2DRef, sparse2super_ref, soft_dice_loss_ref, dice_loss_ref
from lidog_amd.minkunet import make_models
use_bev = "--nobev" not in sys.argv
C = small_batch((0,), n_points=1500)
g = torch.Generator().manual_seed(23)
labels = torch.randint(-1, 7, (C.shape[0],), generator=g); bev_labels = torch.randint(-1, 7, (1, 17, 17), generator=g)
kw = dict(in_channels=1, out_channels=7, D=3, initial_kernel_size=5, decoder_2d_level=["block8"], mapping_bound_2d=5.0)
model = lidog_amd.MinkUNet34BEV(**kw); sd = seeded_state_dict(model, seed=5); model.load_state_dict(sd); model.cuda().train()
st = ME.SparseTensor(coordinates=C.cuda(), features=torch.ones((C.shape[0], 1), device="cuda"))
sem, bev = model(st, is_train=True)
total = 0.5 * SoftDICELoss(ignore_label=-1)(sem.F, labels.cuda())
if use_bev: total = total + 0.5 * DICELoss(ignore_label=-1)(bev["block8"].view(-1, 7), bev_labels.cuda().view(-1))
total.backward()
OME.set_mode("exact")
ref = make_models(OME, Encoder2DRef, lambda x, bound, voxel, pool: sparse2super_ref(x.C, x.F, bound, voxel, pool)).MinkUNet34BEV(**kw)
ref.load_state_dict(sd); ref.train()
rs, rb = ref(OME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1))), is_train=True)
rl = 0.5 * soft_dice_loss_ref(rs.F, labels)
if use_bev: rl = rl + 0.5 * dice_loss_ref(rb["block8"].view(-1, 7), bev_labels.view(-1))
rl.backward()
print("N", C.shape[0], "logit diff", (sem.F.detach().cpu() - rs.F.detach()).abs().max().item(), "loss", float(total), float(rl))
gp = dict(model.named_parameters()); rp = dict(ref.named_parameters())
for n in reversed(list(gp.keys())):
    if gp[n].grad is None or rp[n].grad is None: continue
    a, b = gp[n].grad.cpu(), rp[n].grad
    print("%-48s vec-rel %.2e  norm-rel %.2e  |ref| %.3e" % (n, ((a - b).norm() / (b.norm() + 1e-30)).item(), abs(a.norm() - b.norm()) / (b.norm() + 1e-30), b.norm().item()))
