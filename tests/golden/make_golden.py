"""Generates tests/golden/*.npz by running the REFERENCE's own python in the build container.

Run with:  PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box: only the .npz files travel).

What executes reference code here:
  G1  pixel LUTs            <- MinkUNetBaseBEV.sparse2super            (minkunet_bev.py:169-230)
  G2  sparse2super fwd/bwd  <- the same method, 96 channels, B=5 m
  G3  Encoder2D fwd/bwd     <- utils/models/conv2d.py Encoder2D
  G4  DICE / SoftDICE       <- utils/losses/losses.py
  G5  MinkUNet34BEV         <- utils/models/minkunet_bev.py class, with the CPU oracle
                               (oracle/me_cpu) aliased as MinkowskiEngine, because the real
                               MinkowskiEngine 0.5.4 is not installable here ("parity unpinned"
                               at that boundary, see oracle/me_oracle.c)
  G6  MinkUNet34            <- utils/models/minkunet.py class, same arrangement
  G7  BEV label images      <- PC2ImgConverter.getBEVImageNew (utils/datasets/semantickitti_bev.py:433-464)
  G8  float64 ground truth  <- the G5 / G6 runs repeated with model.double() on the oracle's float64 path
                               (training-mode BatchNorm): gradient vectors + how far the float32 golden run is from them
"""
import hashlib
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
REF = "/root/reference"
sys.path.insert(1, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle.me_cpu as ME  # noqa: E402

ME.install_as_minkowski_engine()
from helpers import seeded_state_dict, small_batch, sha_triples  # noqa: E402
from utils.models.minkunet_bev import MinkUNet34BEV as RefBEV  # noqa: E402  (reference code)
from utils.models.minkunet import MinkUNet34 as RefUNet  # noqa: E402
from utils.models.conv2d import Encoder2D as RefEncoder2D  # noqa: E402
from utils.losses.losses import SoftDICELoss, DICELoss  # noqa: E402

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)


def sha(t):
    return hashlib.sha1(np.ascontiguousarray(t.detach().numpy()).tobytes()).hexdigest()


class _Stub:
    """Carrier for the attributes sparse2super reads from `self`."""

    def __init__(self, bound, pool):
        self.mapping_boundaries = [[-bound, bound], [-bound, bound], [-10, 8]]
        self.pool2D = pool
    filter_bounds = RefBEV.filter_bounds


class _ST:
    def __init__(self, C, F):
        self.C, self.F, self.device = C, F, F.device


def ref_sparse2super(C, F, bound, pool=None):
    stub = _Stub(bound, pool if pool is not None else torch.nn.MaxPool2d(5, 3, 1))
    return RefBEV.sparse2super(stub, _ST(C, F))


def g1_luts():
    out = {}
    for bound in (50.0, 30.0, 5.0):
        H = int(round(2 * bound / 0.05))
        lo, hi = -int(H * 0.6), int(H * 0.6)
        cs = np.arange(lo, hi, dtype=np.int32)
        lut_x = np.full(cs.shape, -1, np.int32)
        lut_y = np.full(cs.shape, -1, np.int32)
        for axis, lut in ((1, lut_x), (2, lut_y)):
            for parity in (0, 1):
                sel = cs[(cs - lo) % 2 == parity]
                C = torch.zeros((sel.shape[0], 4), dtype=torch.int32)
                C[:, axis] = torch.from_numpy(sel)
                F = torch.arange(1, sel.shape[0] + 1, dtype=torch.float32).view(-1, 1)
                img = ref_sparse2super(C, F, bound, pool=torch.nn.Identity())[0, 0]  # [H,W], value = row id + 1
                py, px = torch.nonzero(img, as_tuple=True)
                ids = img[py, px].long() - 1
                lut[sel[ids.numpy()] - lo] = (px if axis == 1 else py).numpy()
        out[f"lut_x_{int(bound)}"] = lut_x
        out[f"lut_y_{int(bound)}"] = lut_y
        out[f"lut_lo_{int(bound)}"] = np.int32(lo)
    np.savez_compressed(os.path.join(HERE, "g1_bev_luts.npz"), **out)
    print("G1", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.shape})


def g2_sparse2super():
    C = small_batch((3, 4), n_points=1500)
    g = torch.Generator().manual_seed(11)
    F = torch.rand((C.shape[0], 96), generator=g)
    F = torch.where(torch.rand(F.shape, generator=g) < 0.3, torch.zeros_like(F), F)  # ReLU-like zeros
    F.requires_grad_(True)
    out = ref_sparse2super(C, F, 5.0)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    idx = torch.randint(0, out.numel(), (4096,), generator=g)
    np.savez_compressed(os.path.join(HERE, "g2_sparse2super.npz"), coords=C.numpy(), seed=11,
                        out_shape=np.array(out.shape), out_sha1=sha(out), out_sample_idx=idx.numpy(),
                        out_sample=out.detach().flatten()[idx].numpy(), out_sum=out.detach().double().sum().numpy(),
                        nnz=int((out != 0).sum()), gin_sha1=sha(F.grad), gin_rowsum=F.grad.sum(dim=1).numpy(),
                        gin_abs_sum=F.grad.abs().double().sum().numpy())
    print("G2", tuple(out.shape), "nnz", int((out != 0).sum()))


def g3_encoder2d():
    enc = RefEncoder2D(96, n_classes=7)
    enc.load_state_dict(seeded_state_dict(enc, seed=3))
    enc.train()
    g = torch.Generator().manual_seed(13)
    x = torch.rand((2, 96, 66, 66), generator=g)
    x = torch.where(torch.rand(x.shape, generator=g) < 0.8, torch.zeros_like(x), x).requires_grad_(True)
    y = enc(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    grads = {f"grad/{n}": p.grad.numpy() for n, p in enc.named_parameters() if p.numel() < 4096}
    norms = {f"gnorm/{n}": p.grad.norm().numpy() for n, p in enc.named_parameters()}
    sd = enc.state_dict()
    np.savez_compressed(os.path.join(HERE, "g3_encoder2d.npz"), y=y.detach().numpy(),
                        gx_rowsum=x.grad.sum(dim=(2, 3)).numpy(), gx_norm=x.grad.norm().numpy(),
                        rm1=sd["down1.maxpool_conv.0.double_conv.1.running_mean"].numpy(),
                        rv1=sd["down1.maxpool_conv.0.double_conv.1.running_var"].numpy(),
                        keys=np.array(list(sd.keys())), **grads, **norms)
    print("G3", tuple(y.shape))


def g4_losses():
    g = torch.Generator().manual_seed(17)
    logits = torch.randn((1024, 7), generator=g).requires_grad_(True)
    labels = torch.randint(-1, 7, (1024,), generator=g)
    out = {"logits": logits.detach().numpy(), "labels": labels.numpy()}
    l1 = SoftDICELoss(ignore_label=-1)(logits, labels)
    l1.backward()
    out["soft_dice"], out["soft_dice_grad"] = l1.detach().numpy(), logits.grad.clone().numpy()
    logits.grad = None
    l2 = DICELoss(ignore_label=-1)(logits, labels)
    l2.backward()
    out["dice"], out["dice_grad"] = l2.detach().numpy(), logits.grad.clone().numpy()
    # the BEV path: NCHW logits through .view(-1, 7) (trainer_lighting_2d.py:181-182)
    bev = torch.randn((2, 7, 16, 16), generator=g).requires_grad_(True)
    bl = torch.randint(-1, 7, (2, 16, 16), generator=g)
    l3 = DICELoss(ignore_label=-1)(bev.view(-1, 7).cpu(), bl.view(-1).cpu())
    l3.backward()
    out.update(bev=bev.detach().numpy(), bev_labels=bl.numpy(), bev_dice=l3.detach().numpy(),
               bev_dice_grad=bev.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "g4_losses.npz"), **out)
    print("G4", float(l1), float(l2), float(l3))


def _kmap_fingerprints(cm):
    fp = {}
    for (s_in, s_out, k, d), (k_off, pin, pout, _) in sorted(cm.kmaps.items()):
        fp[f"kmap_{s_in}_{s_out}_{k}"] = np.array([sha_triples(k_off, pin, pout), str(int(k_off[-1]))])
    return fp


G5_GRAD_VECTORS = ["conv1p1s2.kernel", "block1.0.conv1.kernel", "block1.1.norm2.bn.weight", "conv2p2s2.kernel",
                   "block2.0.downsample.0.kernel", "block2.2.conv2.kernel", "conv3p4s2.kernel", "block3.0.norm1.bn.bias",
                   "block3.0.downsample.0.kernel", "conv4p8s2.kernel", "block4.0.downsample.0.kernel",
                   "block4.5.norm2.bn.weight", "bntr4.bn.weight", "block5.0.downsample.0.kernel", "convtr6p4s2.kernel",
                   "block7.0.downsample.0.kernel", "convtr7p2s2.kernel", "block8.1.conv2.kernel",
                   "encoders2d.block8.out_conv.conv.weight"]


def g5_full_model():
    C = small_batch((0, 1))
    N = C.shape[0]
    g = torch.Generator().manual_seed(23)
    labels = torch.randint(-1, 7, (N,), generator=g)
    bev_labels = torch.randint(-1, 7, (2, 17, 17), generator=g)
    model = RefBEV(in_channels=1, out_channels=7, D=3, initial_kernel_size=5, decoder_2d_level=["block8"],
                   mapping_bound_2d=5.0)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    feats = torch.ones((N, 1))
    out = {}
    # validation path first (is_train=False: no BEV head, seeded running statistics; minkunet_bev.py:376-393)
    model.eval()
    with torch.no_grad():
        sem0, none0 = model(ME.SparseTensor(coordinates=C, features=feats), is_train=False)
    assert none0 is None
    out["eval_logits_initial"] = sem0.F.numpy().copy()
    sem_c, bev_c = SoftDICELoss(ignore_label=-1), DICELoss(ignore_label=-1)
    # Full backward chain with BatchNorm in evaluation mode (running statistics), BEV head included: without batch
    # statistics nothing amplifies summation-order noise, so these gradient vectors can be compared tightly (the
    # training-mode gradients below move by 1 - cos = 2e-3..4e-3 when the SAME oracle merely runs on 8 threads)
    sem_e, bev_e = model(ME.SparseTensor(coordinates=C, features=feats), is_train=True)
    loss_e = 0.5 * sem_c(sem_e.F, labels).cpu() + 0.5 * bev_c(bev_e["block8"].view(-1, 7).cpu(), bev_labels.view(-1).cpu())
    model.zero_grad()
    loss_e.backward()
    out["eval_loss"] = loss_e.detach().numpy()
    for n in G5_GRAD_VECTORS:
        gvec = dict(model.named_parameters())[n].grad.numpy()
        scale = float(np.abs(gvec).max())
        out[f"grad16_eval/{n}"] = (gvec / scale).astype(np.float16)
        out[f"grad16scale_eval/{n}"] = np.float32(scale)
    model.zero_grad()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    losses = []
    for step in range(3):
        st = ME.SparseTensor(coordinates=C, features=feats)
        sem, bev = model(st, is_train=True)
        l_bev = bev_c(bev["block8"].view(-1, 7).cpu(), bev_labels.view(-1).cpu())
        l_sem = sem_c(sem.F, labels).cpu()
        total = 0.5 * l_sem + 0.5 * l_bev
        opt.zero_grad()
        total.backward()
        if step == 0:
            cm = st.coordinate_manager
            out.update(logits=sem.F.detach().numpy(), bev_logits=bev["block8"].detach().numpy(),
                       n_vox=np.array([cm.maps[s].shape[0] for s in (1, 2, 4, 8, 16)]),
                       **_kmap_fingerprints(cm))
            for s in (2, 4, 8, 16):
                out[f"coords_s{s}_sha1"] = np.array(sha(cm.maps[s]))
            for n, p in model.named_parameters():
                out[f"gnorm/{n}"] = p.grad.norm().numpy()
            out["grad/final.kernel"] = model.final.kernel.grad.numpy()
            out["grad/final.bias"] = model.final.bias.grad.numpy()
            out["grad/conv0p1s1.kernel"] = model.conv0p1s1.kernel.grad.numpy()
            out["grad/bn0.bn.weight"] = model.bn0.bn.weight.grad.numpy()
            # full gradient VECTORS of parameters spread over the depth of the network (cosine test of the whole
            # backward chain), normalised to max |g| = 1 and stored as float16 (relative 5e-4 per element)
            for n in G5_GRAD_VECTORS:
                gvec = dict(model.named_parameters())[n].grad.numpy()
                scale = float(np.abs(gvec).max())
                out[f"grad16/{n}"] = (gvec / scale).astype(np.float16)
                out[f"grad16scale/{n}"] = np.float32(scale)
            out["bn0_running_mean"] = model.bn0.bn.running_mean.numpy().copy()
            out["bn0_running_var"] = model.bn0.bn.running_var.numpy().copy()
        losses.append([float(l_sem), float(l_bev), float(total)])
        opt.step()
    # eval-mode forward (validation path, is_train=False -> no BEV head, running stats)
    model.eval()
    with torch.no_grad():
        sem, none = model(ME.SparseTensor(coordinates=C, features=feats), is_train=False)
    assert none is None
    out["eval_logits_after3"] = sem.F.numpy()
    np.savez_compressed(os.path.join(HERE, "g5_minkunet34bev.npz"), coords=C.numpy(), labels=labels.numpy(),
                        bev_labels=bev_labels.numpy(), losses=np.array(losses), keys=np.array(list(model.state_dict().keys())),
                        **out)
    print("G5", N, out["n_vox"], losses)


def g6_unet():
    C = small_batch((7,), n_points=2000)
    N = C.shape[0]
    g = torch.Generator().manual_seed(29)
    labels = torch.randint(-1, 7, (N,), generator=g)
    model = RefUNet(in_channels=1, out_channels=7, D=3)
    model.load_state_dict(seeded_state_dict(model, seed=7))
    model.train()
    sem = model(ME.SparseTensor(coordinates=C, features=torch.ones((N, 1))), is_seg=True)
    loss = SoftDICELoss(ignore_label=-1)(sem.F, labels)
    loss.backward()
    out = {f"gnorm/{n}": p.grad.norm().numpy() for n, p in model.named_parameters()}
    np.savez_compressed(os.path.join(HERE, "g6_minkunet34.npz"), coords=C.numpy(), labels=labels.numpy(),
                        logits=sem.F.detach().numpy(), loss=loss.detach().numpy(),
                        keys=np.array(list(model.state_dict().keys())), **out)
    print("G6", N, float(loss))


def g7_bev_labels():
    """PC2ImgConverter.getBEVImageNew (reference code) on voxel coordinates: the BEV label rasteriser (N2).
    torchvision is not installed and the file uses np.int (removed from numpy): both are patched for the import
    only; the method itself is pure numpy."""
    import types
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.Compose = object
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm"))
    if not hasattr(np, "int"):
        np.int = int
    from utils.datasets.semantickitti_bev import PC2ImgConverter
    rng = np.random.default_rng(41)
    out = {}
    for bound, size in ((50.0, 167), (30.0, 100)):
        n = 12000
        vox = np.stack([rng.integers(-1250, 1250, n), rng.integers(-1250, 1250, n), rng.integers(-220, 180, n)], axis=1)
        # include the coordinates that sit on the float32 bound and many collisions per pixel
        vox[:40, 0] = np.array([int(bound / 0.05) - 1, -int(bound / 0.05) + 1] * 20)
        vox = np.unique(vox, axis=0)
        vox = vox[rng.permutation(vox.shape[0])].astype(np.int32)
        labels = rng.integers(-1, 7, vox.shape[0])
        grid = (bound - (-bound)) / size
        conv = PC2ImgConverter(imgChannel=1, xRange=[-bound, bound], yRange=[-bound, bound], zRange=[-10, 8],
                               xGridSize=grid, yGridSize=grid, zGridSize=0.3)
        pts = (vox * 0.05).astype(np.float32)
        img, idx = conv.getBEVImageNew(pts, labels)
        tag = str(int(bound))
        out.update({f"vox_{tag}": vox, f"labels_{tag}": labels, f"img_{tag}": img.astype(np.int32),
                    f"idx_{tag}": idx.astype(np.int32)})
        print("G7", bound, img.shape, int((img >= 0).sum()))
    np.savez_compressed(os.path.join(HERE, "g7_bev_labels.npz"), **out)


G8_VECTORS = ["conv1p1s2.kernel", "bn1.bn.weight", "block1.0.conv1.kernel", "block1.1.norm2.bn.weight",
              "block1.1.norm2.bn.bias", "conv2p2s2.kernel", "block2.0.downsample.0.kernel", "block2.0.norm1.bn.weight",
              "conv3p4s2.kernel", "block3.0.norm1.bn.bias", "block3.0.downsample.0.kernel", "conv4p8s2.kernel",
              "block4.0.downsample.0.kernel", "block4.5.norm2.bn.weight", "bntr4.bn.weight", "bntr4.bn.bias",
              "block5.0.downsample.0.kernel", "convtr6p4s2.kernel", "block7.0.downsample.0.kernel", "convtr7p2s2.kernel",
              "block8.0.norm1.bn.weight", "block8.1.norm2.bn.weight", "block8.1.norm2.bn.bias", "final.kernel"]


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def g8_float64():
    """Ground truth for the training-mode gradient checks: the reference's classes on the oracle in float64
    (convolutions, BatchNorm, BEV head, losses all in double).  Stored per vector: the float64 gradient (as float32)
    and the relative L2 distance of the float32 golden run (same code, float32, one thread) from it -- the yardstick
    the HIP path is held to (tests/test_gpu_model.py)."""
    out = {}

    def run_bev(dt):
        C = small_batch((0, 1))
        N = C.shape[0]
        g = torch.Generator().manual_seed(23)
        labels = torch.randint(-1, 7, (N,), generator=g)
        bev_labels = torch.randint(-1, 7, (2, 17, 17), generator=g)
        model = RefBEV(in_channels=1, out_channels=7, D=3, initial_kernel_size=5, decoder_2d_level=["block8"],
                       mapping_bound_2d=5.0)
        model.load_state_dict(seeded_state_dict(model, seed=5))
        if dt == torch.float64:
            model.double()
            zeros = torch.zeros

            def s2s(x, **kw):   # the reference method allocates its image with torch.zeros (float32): keep float64
                torch.zeros = lambda *a, **k: zeros(*a, **({**k, "dtype": torch.float64} if "dtype" not in k else k))
                try:
                    return RefBEV.sparse2super(model, x, **kw)
                finally:
                    torch.zeros = zeros
            model.sparse2super = s2s
        model.train()
        sem, bev = model(ME.SparseTensor(coordinates=C, features=torch.ones((N, 1), dtype=dt)), is_train=True)
        loss = 0.5 * SoftDICELoss(ignore_label=-1)(sem.F, labels).cpu() + \
            0.5 * DICELoss(ignore_label=-1)(bev["block8"].view(-1, 7).cpu(), bev_labels.view(-1).cpu())
        loss.backward()
        return model, sem, bev, loss

    def run_unet(dt):
        C = small_batch((7,), n_points=2000)
        N = C.shape[0]
        g = torch.Generator().manual_seed(29)
        labels = torch.randint(-1, 7, (N,), generator=g)
        model = RefUNet(in_channels=1, out_channels=7, D=3)
        model.load_state_dict(seeded_state_dict(model, seed=7))
        if dt == torch.float64:
            model.double()
        model.train()
        sem = model(ME.SparseTensor(coordinates=C, features=torch.ones((N, 1), dtype=dt)), is_seg=True)
        loss = SoftDICELoss(ignore_label=-1)(sem.F, labels)
        loss.backward()
        return model, sem, None, loss

    for tag, run, names in (("g5", run_bev, G8_VECTORS + ["encoders2d.block8.out_conv.conv.weight"]),
                            ("g6", run_unet, G8_VECTORS)):
        m64, sem64, bev64, loss64 = run(torch.float64)
        m32, sem32, bev32, loss32 = run(torch.float32)
        p64, p32 = dict(m64.named_parameters()), dict(m32.named_parameters())
        out[f"{tag}/logits64"] = sem64.F.detach().numpy().astype(np.float32)
        out[f"{tag}/logits_err32"] = np.float64((sem64.F.detach() - sem32.F.detach().double()).abs().max())
        out[f"{tag}/loss64"] = np.float64(loss64.detach())
        if bev64 is not None:
            out[f"{tag}/bev_logits64"] = bev64["block8"].detach().numpy().astype(np.float32)
        errs = []
        for n in names:
            out[f"{tag}/grad64/{n}"] = p64[n].grad.numpy().astype(np.float32)
            out[f"{tag}/err32/{n}"] = np.float64(_rel(p32[n].grad.numpy(), p64[n].grad.numpy()))
            errs.append(float(out[f"{tag}/err32/{n}"]))
        # every parameter: the distance of the float32 run from the float64 one (norm-level view of the whole chain)
        out[f"{tag}/all_names"] = np.array(list(p64))
        out[f"{tag}/all_err32"] = np.array([_rel(p32[n].grad.numpy(), p64[n].grad.numpy()) for n in p64])
        out[f"{tag}/all_gnorm64"] = np.array([float(p64[n].grad.norm()) for n in p64])
        print("G8", tag, "float32 golden vs float64: logits", float(out[f"{tag}/logits_err32"]), "gradient vectors rel L2 min / median / max",
              min(errs), float(np.median(errs)), max(errs))
    np.savez_compressed(os.path.join(HERE, "g8_float64.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    fns = dict(g1=g1_luts, g2=g2_sparse2super, g3=g3_encoder2d, g4=g4_losses, g5=g5_full_model, g6=g6_unet,
               g7=g7_bev_labels, g8=g8_float64)
    for w in which:
        fns[w]()
