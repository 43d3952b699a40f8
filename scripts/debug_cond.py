"""GPU self-spread of gradients under a 1-ulp parameter perturbation (conditioning of the G5 model)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import GOLDEN, seeded_state_dict
import lidog_amd, lidog_amd.me as ME
from lidog_amd.losses import SoftDICELoss, DICELoss
g5 = np.load(f"{GOLDEN}/g5_minkunet34bev.npz")
C = torch.from_numpy(g5["coords"]).cuda(); labels = torch.from_numpy(g5["labels"]).cuda(); bev_labels = torch.from_numpy(g5["bev_labels"]).cuda()
def run(perturb):
    model = lidog_amd.MinkUNet34BEV(in_channels=1, out_channels=7, D=3, initial_kernel_size=5, decoder_2d_level=["block8"], mapping_bound_2d=5.0)
    sd = seeded_state_dict(model, seed=5)
    if perturb:
        g = torch.Generator().manual_seed(99)
        sd = {k: (v * (1 + perturb * torch.randn(v.shape, generator=g)) if v.dtype == torch.float32 else v) for k, v in sd.items()}
    model.load_state_dict(sd); model.cuda().train()
    st = ME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1), device="cuda"))
    sem, bev = model(st, is_train=True)
    total = 0.5 * SoftDICELoss(ignore_label=-1)(sem.F, labels) + 0.5 * DICELoss(ignore_label=-1)(bev["block8"].view(-1, 7), bev_labels.view(-1))
    total.backward()
    return sem.F.detach().cpu(), {n: p.grad.detach().cpu() for n, p in model.named_parameters()}
l0, g0 = run(0.0)
l0b, g0b = run(0.0)
print("rerun identical:", (l0 - l0b).abs().max().item(), max((g0[n] - g0b[n]).abs().max().item() for n in g0))
for eps in (1e-7, 1e-6):
    l1, g1 = run(eps)
    rel = sorted(((abs(float(g1[n].norm()) - float(g0[n].norm())) / float(g0[n].norm()), n) for n in g0), reverse=True)
    print(f"perturb {eps}: logit diff {(l1 - l0).abs().max().item():.2e}; grad-norm rel diff max {rel[0][0]:.2e} ({rel[0][1]}) median {np.median([r[0] for r in rel]):.2e}")
refrel = sorted(((abs(float(g0[n].norm()) - float(g5['gnorm/' + n])) / float(g5['gnorm/' + n]), n) for n in g0), reverse=True)
print(f"vs golden: max {refrel[0][0]:.2e} ({refrel[0][1]}) median {np.median([r[0] for r in refrel]):.2e}")
