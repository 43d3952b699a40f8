"""The reference's literal call order on the HIP operators (the zero-edit alias route).

`utils/models/minkunet_bev.py:302-374` walks the network as separate modules: `conv -> bn -> relu` (MinkowskiReLU
inplace), `ME.cat`, BasicBlocks made by `_make_layer`, `final`.  lidog_amd.minkunet restructures the same graph
around fused conv+BN kernels and the trunk executor; this file restates the reference's forward (the call ORDER, written
here from the cited lines -- not the reference file) over `lidog_amd.me` and checks that the module-by-module route
gives the golden logits of the reference classes (G5, 1e-4) and the fused wiring's results (logits 1e-6, gradient
cosines 1 - 1e-6).  Two block classes: `ME.modules.resnet_block.BasicBlock` as the alias hands it out, and a literal
block (conv1, norm1, relu, conv2, norm2, `out += residual`, relu: utils/models/resnet_block.py:8-56) that exercises
`SparseTensor.__iadd__` and the in-place ReLU."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import GOLDEN, seeded_state_dict

pytestmark = pytest.mark.gpu


def _literal_model(block_kind):
    import lidog_amd.me as ME
    from lidog_amd import bev

    class LiteralBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
            super().__init__()
            self.conv1 = ME.MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                                 dimension=dimension)
            self.norm1 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
            self.conv2 = ME.MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                                 dimension=dimension)
            self.norm2 = ME.MinkowskiBatchNorm(planes, momentum=bn_momentum)
            self.relu = ME.MinkowskiReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            residual = x
            out = self.conv1(x)
            out = self.norm1(out)
            out = self.relu(out)
            out = self.conv2(out)
            out = self.norm2(out)
            if self.downsample is not None:
                residual = self.downsample(x)
            out += residual
            out = self.relu(out)
            return out

    Block = ME.modules.resnet_block.BasicBlock if block_kind == "me_block" else LiteralBlock

    class Literal(nn.Module):
        """layers: minkunet_bev.py:44-156; forward: :302-399 (is_seg=True, no binary head)"""
        PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
        LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)

        def __init__(self, in_channels, out_channels, D, bound):
            super().__init__()
            self.D, self.bound = D, bound
            self.inplanes = 32
            P, L = self.PLANES, self.LAYERS
            conv, convtr, bn = ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose, ME.MinkowskiBatchNorm
            self.conv0p1s1 = conv(in_channels, self.inplanes, kernel_size=5, dimension=D)
            self.bn0 = bn(self.inplanes)
            self.conv1p1s2 = conv(self.inplanes, self.inplanes, kernel_size=2, stride=2, dimension=D)
            self.bn1 = bn(self.inplanes)
            self.block1 = self._make_layer(Block, P[0], L[0])
            self.conv2p2s2 = conv(self.inplanes, self.inplanes, kernel_size=2, stride=2, dimension=D)
            self.bn2 = bn(self.inplanes)
            self.block2 = self._make_layer(Block, P[1], L[1])
            self.conv3p4s2 = conv(self.inplanes, self.inplanes, kernel_size=2, stride=2, dimension=D)
            self.bn3 = bn(self.inplanes)
            self.block3 = self._make_layer(Block, P[2], L[2])
            self.conv4p8s2 = conv(self.inplanes, self.inplanes, kernel_size=2, stride=2, dimension=D)
            self.bn4 = bn(self.inplanes)
            self.block4 = self._make_layer(Block, P[3], L[3])
            self.convtr4p16s2 = convtr(self.inplanes, P[4], kernel_size=2, stride=2, dimension=D)
            self.bntr4 = bn(P[4])
            self.inplanes = P[4] + P[2]
            self.block5 = self._make_layer(Block, P[4], L[4])
            self.convtr5p8s2 = convtr(self.inplanes, P[5], kernel_size=2, stride=2, dimension=D)
            self.bntr5 = bn(P[5])
            self.inplanes = P[5] + P[1]
            self.block6 = self._make_layer(Block, P[5], L[5])
            self.convtr6p4s2 = convtr(self.inplanes, P[6], kernel_size=2, stride=2, dimension=D)
            self.bntr6 = bn(P[6])
            self.inplanes = P[6] + P[0]
            self.block7 = self._make_layer(Block, P[6], L[6])
            self.convtr7p2s2 = convtr(self.inplanes, P[7], kernel_size=2, stride=2, dimension=D)
            self.bntr7 = bn(P[7])
            self.inplanes = P[7] + 32
            self.block8 = self._make_layer(Block, P[7], L[7])
            self.final = conv(P[7], out_channels, kernel_size=1, bias=True, dimension=D)
            self.relu = ME.MinkowskiReLU(inplace=True)
            self.dropout = ME.MinkowskiDropout(p=0.5)
            self.encoders2d = nn.ModuleDict({"block8": bev.Encoder2D(96, n_classes=out_channels)})

        def _make_layer(self, block, planes, blocks):
            downsample = None
            if self.inplanes != planes * block.expansion:
                downsample = nn.Sequential(
                    ME.MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1, stride=1,
                                            dimension=self.D),
                    ME.MinkowskiBatchNorm(planes * block.expansion))
            layers = [block(self.inplanes, planes, stride=1, dilation=1, downsample=downsample, dimension=self.D)]
            self.inplanes = planes * block.expansion
            for _ in range(1, blocks):
                layers.append(block(self.inplanes, planes, stride=1, dilation=1, dimension=self.D))
            return nn.Sequential(*layers)

        def forward(self, x, is_train=False):
            out = self.conv0p1s1(x)
            out = self.bn0(out)
            out_p1 = self.relu(out)
            out = self.conv1p1s2(out_p1)
            out = self.bn1(out)
            out = self.relu(out)
            out_b1p2 = self.block1(out)
            out = self.conv2p2s2(out_b1p2)
            out = self.bn2(out)
            out = self.relu(out)
            out_b2p4 = self.block2(out)
            out = self.conv3p4s2(out_b2p4)
            out = self.bn3(out)
            out = self.relu(out)
            out_b3p8 = self.block3(out)
            out = self.conv4p8s2(out_b3p8)
            out = self.bn4(out)
            out = self.relu(out)
            out_bottle = self.block4(out)
            out = self.convtr4p16s2(out_bottle)
            out = self.bntr4(out)
            out = self.relu(out)
            out = ME.cat(out, out_b3p8)
            out_block5 = self.block5(out)
            out = self.convtr5p8s2(out_block5)
            out = self.bntr5(out)
            out = self.relu(out)
            out = ME.cat(out, out_b2p4)
            out_block6 = self.block6(out)
            out = self.convtr6p4s2(out_block6)
            out = self.bntr6(out)
            out = self.relu(out)
            out = ME.cat(out, out_b1p2)
            out_block7 = self.block7(out)
            out = self.convtr7p2s2(out_block7)
            out = self.bntr7(out)
            out = self.relu(out)
            out = ME.cat(out, out_p1)
            out_block8 = self.block8(out)
            img_pred = None
            if is_train:
                img = bev.sparse2super(out_block8, bound=self.bound, voxel=0.05, pool=(5, 3, 1))
                img_pred = {"block8": self.encoders2d["block8"](img)}
            return self.final(out_block8), img_pred

    return Literal(1, 7, 3, 5.0)


def _step(model, C, labels, bev_labels):
    import lidog_amd.me as ME
    from lidog_amd.losses import SoftDICELoss, DICELoss
    model.zero_grad()
    sem, bev = model(ME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1), device="cuda")), is_train=True)
    loss = 0.5 * SoftDICELoss(ignore_label=-1)(sem.F, labels) + \
        0.5 * DICELoss(ignore_label=-1)(bev["block8"].view(-1, 7), bev_labels.view(-1))
    loss.backward()
    return sem.F.detach().clone(), bev["block8"].detach().clone(), float(loss.detach()), \
        {n: p.grad.detach().clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("block_kind", ["me_block", "literal_block"])
def test_reference_call_order_on_hip_operators(block_kind):
    import lidog_amd
    from lidog_amd import trunk
    g5 = np.load(f"{GOLDEN}/g5_minkunet34bev.npz")
    C = torch.from_numpy(g5["coords"]).cuda()
    labels = torch.from_numpy(g5["labels"]).cuda()
    bev_labels = torch.from_numpy(g5["bev_labels"]).cuda()
    lit = _literal_model(block_kind)
    assert list(lit.state_dict().keys()) == list(g5["keys"]), "state_dict keys differ from the reference model"
    sd = seeded_state_dict(lit, seed=5)
    lit.load_state_dict(sd)
    lit.cuda().train()
    sem_l, bev_l, loss_l, grads_l = _step(lit, C, labels, bev_labels)
    # against the golden run of the reference's own classes
    assert (sem_l.cpu() - torch.from_numpy(g5["logits"])).abs().max().item() <= 1e-4
    assert (bev_l.cpu() - torch.from_numpy(g5["bev_logits"])).abs().max().item() <= 1e-4
    assert abs(loss_l - float(g5["losses"][0][2])) <= 1e-5
    # against the fused wiring of lidog_amd.minkunet, operator path and executor
    for executor in (False, True):
        trunk.set_enabled(executor)
        try:
            fused = lidog_amd.MinkUNet34BEV(in_channels=1, out_channels=7, D=3, initial_kernel_size=5,
                                            decoder_2d_level=["block8"], mapping_bound_2d=5.0)
            fused.load_state_dict(sd)
            fused.cuda().train()
            sem_f, bev_f, loss_f, grads_f = _step(fused, C, labels, bev_labels)
        finally:
            trunk.set_enabled(True)
        scale = float(sem_f.abs().max())
        assert (sem_l - sem_f).abs().max().item() <= 1e-6 * max(1.0, scale), (sem_l - sem_f).abs().max().item()
        assert (bev_l - bev_f).abs().max().item() <= 1e-6 * max(1.0, float(bev_f.abs().max()))
        assert abs(loss_l - loss_f) <= 1e-6
        worst = 0.0
        for n, g in grads_f.items():
            a, b = grads_l[n].double().flatten(), g.double().flatten()
            worst = max(worst, 1 - float(torch.dot(a, b) / (a.norm() * b.norm())))
        print(f"{block_kind} vs fused ({'executor' if executor else 'operator path'}): worst 1 - cos = {worst:.2e}")
        assert worst <= 1e-6, worst
    # running statistics moved the same way
    torch.testing.assert_close(lit.bn0.bn.running_mean, fused.bn0.bn.running_mean, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(lit.block8[1].norm2.bn.running_var, fused.block8[1].norm2.bn.running_var, rtol=1e-6, atol=1e-7)
