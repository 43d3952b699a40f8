import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import GOLDEN, seeded_state_dict
import lidog_amd, lidog_amd.me as ME
from lidog_amd.losses import SoftDICELoss, DICELoss
g5 = np.load(f"{GOLDEN}/g5_minkunet34bev.npz")
C = torch.from_numpy(g5["coords"]).cuda(); labels = torch.from_numpy(g5["labels"]).cuda(); bev_labels = torch.from_numpy(g5["bev_labels"]).cuda()
model = lidog_amd.MinkUNet34BEV(in_channels=1, out_channels=7, D=3, initial_kernel_size=5, decoder_2d_level=["block8"], mapping_bound_2d=5.0)
model.load_state_dict(seeded_state_dict(model, seed=5)); model.cuda().train()
st = ME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1), device="cuda"))
sem, bev = model(st, is_train=True)
total = 0.5 * SoftDICELoss(ignore_label=-1)(sem.F, labels) + 0.5 * DICELoss(ignore_label=-1)(bev["block8"].view(-1, 7), bev_labels.view(-1))
total.backward()
print("logit diff", (sem.F.detach().cpu() - torch.from_numpy(g5["logits"])).abs().max().item())
rows = []
for n, p in model.named_parameters():
    ref = float(g5[f"gnorm/{n}"]); got = float(p.grad.norm())
    rows.append((abs(got - ref) / (ref + 1e-12), n, got, ref))
rows.sort(reverse=True)
for r in rows[:25]: print("%.2e  %-45s %.6e %.6e" % r)
print("median rel", np.median([r[0] for r in rows]))
for n in ("final.kernel", "final.bias", "conv0p1s1.kernel", "bn0.bn.weight"):
    got = dict(model.named_parameters())[n].grad.cpu(); ref = torch.from_numpy(g5[f"grad/{n}"])
    print(n, "max abs diff", (got - ref).abs().max().item(), "ref max", ref.abs().max().item())
