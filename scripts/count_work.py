"""Recomputes voxel counts N_s, rule-book sizes P and the algorithmic FLOPs / bytes per scan of the
MinkUNet34(+BEV) forward from a synthetic scan (SURVEY.md 8(d) formulas), using the CPU oracle's maps.

    python scripts/count_work.py [config] [seed] [mix3d]      (mix3d: the union of scans 2 seed and 2 seed + 1)
FLOPs_l = 2 P_l Cin Cout;  bytes_l = 4 (N_in Cin + N_out Cout) + 4 K Cin Cout + 8 P_l;  BN(+ReLU) = 12 N C.
"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle.me_cpu as OME  # noqa: E402
from lidog_amd import synth  # noqa: E402
from lidog_amd.minkunet import make_models  # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "kitti120k"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    mix = len(sys.argv) > 3 and sys.argv[3] == "mix3d"
    vox, _ = (synth.mix3d_voxels if mix else synth.scan_voxels)(seed, config)
    C = torch.from_numpy(np.concatenate([np.zeros((vox.shape[0], 1), np.int32), vox], axis=1))
    st = OME.SparseTensor(coordinates=C, features=torch.ones((C.shape[0], 1)))
    cm = st.coordinate_manager
    prev = 1
    for s in (2, 4, 8, 16):
        cm.stride(prev, s)
        prev = s
    N = {s: cm.maps[s].shape[0] for s in (1, 2, 4, 8, 16)}
    model = make_models(OME).MinkUNet34(1, 7, 3)
    flops = bytes_ = bn_bytes = pairs = 0
    rows = []

    def visit(mod, s_in):
        nonlocal flops, bytes_, pairs
        K = mod.kernel_volume
        if K == 1:
            s_out, P = s_in, N[s_in]
        elif isinstance(mod, OME.MinkowskiConvolutionTranspose):
            s_out = s_in // 2
            P = int(cm.kernel_map(s_out, s_in, mod.kernel_size)[0][-1])
        else:
            s_out = s_in * mod.stride
            P = int(cm.kernel_map(s_in, s_out, mod.kernel_size)[0][-1])
        f = 2 * P * mod.in_channels * mod.out_channels
        b = 4 * (N[s_in] * mod.in_channels + N[s_out] * mod.out_channels) + 4 * K * mod.in_channels * mod.out_channels + 8 * P
        flops += f
        bytes_ += b
        pairs += P
        return s_out, P, f, b

    # walk the modules in forward order with their tensor strides
    order = [("conv0p1s1", 1)]
    s = 1
    for i, blk in zip((1, 2, 3, 4), ("block1", "block2", "block3", "block4")):
        order.append((f"conv{i}p{s}s2", s))
        s *= 2
        order.append((blk, s))
    for j, blk in zip((4, 5, 6, 7), ("block5", "block6", "block7", "block8")):
        order.append((f"convtr{j}p{s}s2", s))
        s //= 2
        order.append((blk, s))
    order.append(("final", 1))
    for name, s_in in order:
        mod = getattr(model, name)
        convs = [m for m in mod.modules() if isinstance(m, (OME.MinkowskiConvolution, OME.MinkowskiConvolutionTranspose))]
        for m in convs:
            s_out, P, f, b = visit(m, s_in)
            rows.append((name, m.kernel_volume, m.in_channels, m.out_channels, s_in, s_out, P, f / 1e9, b / 1e6))
        for m in mod.modules():
            if isinstance(m, OME.MinkowskiBatchNorm):
                pass
    for m in model.modules():
        if isinstance(m, OME.MinkowskiBatchNorm):
            pass
    # BN traffic: every BN sits on the output of a conv at that conv's output stride
    bn_bytes = sum(12 * N[r[5]] * r[3] for r in rows if r[0] != "final")
    print(f"config {config} seed {seed}: N = {[N[s] for s in (1, 2, 4, 8, 16)]}")
    kinds = {}
    for (s_in, s_out, k, d), v in sorted(cm.kmaps.items()):
        kinds[(s_in, s_out, k)] = int(v[0][-1])
    print("kernel maps (s_in, s_out, k) -> pairs:", kinds)
    print(f"convs {len(rows)}  sum P = {pairs / 1e6:.2f} M  sparse fwd = {flops / 1e9:.1f} GFLOP  conv bytes = {bytes_ / 1e9:.2f} GB  "
          f"BN bytes = {bn_bytes / 1e9:.2f} GB  -> {(bytes_ + bn_bytes) / 1e9:.2f} GB compulsory")
    H = 666
    bev_flops = 2 * (333 * 333 * 256 * 96 * 9 + 167 * 167 * 256 * 256 * 9 + 167 * 167 * 7 * 256)
    bev_bytes = 4 * (N[1] * 96 + 96 * H * H) + 4 * (96 * H * H + 3 * 256 * 333 * 333 + 3 * 256 * 167 * 167 + 7 * 167 * 167) \
        + 4 * (256 * 96 * 9 + 256 * 256 * 9)
    print(f"BEV head fwd (B=50): {bev_flops / 1e9:.1f} GFLOP, {bev_bytes / 1e9:.2f} GB")
    print(f"per scan forward: {(flops + bev_flops) / 1e9:.1f} GFLOP, {(bytes_ + bn_bytes + bev_bytes) / 1e9:.2f} GB; "
          f"training step ~3x: {3 * (flops + bev_flops) / 1e9:.0f} GFLOP, {3 * (bytes_ + bn_bytes + bev_bytes) / 1e9:.1f} GB")
    if "-v" in sys.argv:
        for r in rows:
            print("%-14s K=%3d %3d->%3d s%2d->%2d P=%8d %7.2f GF %7.1f MB" % r)


if __name__ == "__main__":
    main()
