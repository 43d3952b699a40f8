"""One weight-gradient layer of the bench workload, a few launches (for rocprofv3 --pmc):  S CIN COUT from the environment"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import lidog_amd.me as ME
from lidog_amd import synth, _lib
from lidog_amd._lib import call, ptr
b = synth.make_batch(range(4), "kitti120k", "cuda")
st = ME.SparseTensor(coordinates=b["coords_int"], features=b["source_features0"])
cm = st.coordinate_manager
prev = 1
for s in (2, 4, 8, 16):
    cm.stride(prev, s); prev = s
s, Cin, Cout = int(os.environ.get("S", 1)), int(os.environ.get("CIN", 96)), int(os.environ.get("COUT", 96))
m = cm.kernel_map(s, s, 3)
x = torch.randn(m.n_in, Cin, device="cuda"); g = torch.randn(m.n_out, Cout, device="cuda")
items, n_items, item_off = ME._wgrad_items(m, Cin, Cout)
BLOCK = int(os.environ.get("BLOCK", 0))
if BLOCK:   # experiment: items cut by output-row blocks, the 27 items of a block next to each other on one XCD
    import numpy as np
    k_off = np.asarray(m.k_off_host, dtype=np.int64)
    po = m.pair_out.cpu().numpy()
    nb = (m.n_out + BLOCK - 1) // BLOCK
    edges = np.arange(nb + 1, dtype=np.int64) * BLOCK
    rows = []
    for k in range(m.K):
        seg = po[k_off[k]:k_off[k + 1]]
        cuts = k_off[k] + np.searchsorted(seg, edges, side="left")
        for b in range(nb):
            rows.append((k, cuts[b], cuts[b + 1], b))
    rows = np.array(rows, dtype=np.int64)                      # ordered by k, then block
    n_items = len(rows)
    ids = np.arange(n_items).reshape(m.K, nb)                  # item id of (k, block)
    order = np.empty(n_items, dtype=np.int64)
    pos = 0
    G = m.K
    for s0 in range(0, nb, 8):                                  # 8 blocks at a time, one per XCD
        blocks = list(range(s0, min(s0 + 8, nb)))
        span = np.full(8 * G, -1, dtype=np.int64)
        for xcd, b in enumerate(blocks):
            for j in range(G):
                span[8 * j + xcd] = ids[j, b]
        span = span[span >= 0]
        order[pos:pos + len(span)] = span
        pos += len(span)
    items_np = np.stack([rows[:, 0], rows[:, 1], rows[:, 2], order]).astype(np.int32)
    items = torch.from_numpy(items_np).cuda()
    item_off = torch.from_numpy((np.arange(m.K + 1) * nb).astype(np.int32)).cuda()
    print(f"row-block items: {nb} blocks of {BLOCK} rows, {n_items} items, longest {int((rows[:,2]-rows[:,1]).max())} pairs")
HYB = int(os.environ.get("HYBRID", 0))
if HYB:   # experiment (round 5): the dense offsets (>= DENSE pairs per row) cut by output-row blocks of HYB rows, the items of one
    # block next to each other on one XCD; the sparse offsets stay equal-length pair ranges
    import numpy as np
    DENSE = float(os.environ.get("DENSE", 0.2))
    k_off = np.asarray(m.k_off_host, dtype=np.int64)
    po = m.pair_out.cpu().numpy()
    cnt = np.diff(k_off)
    dense = [k for k in range(m.K) if cnt[k] >= DENSE * m.n_out]
    nb = (m.n_out + HYB - 1) // HYB
    edges = np.arange(nb + 1, dtype=np.int64) * HYB
    chunk = ME._wgrad_chunk(m.k_off_host, Cin, Cout)
    rows, tag = [], []          # (k, p0, p1), tag = block id for dense items, -1 for sparse ones
    for k in range(m.K):
        if k in dense:
            seg = po[k_off[k]:k_off[k + 1]]
            cuts = k_off[k] + np.searchsorted(seg, edges, side="left")
            for b in range(nb):
                rows.append((k, cuts[b], cuts[b + 1])); tag.append(b)
        else:
            pp = k_off[k]
            while pp < k_off[k + 1]:
                qq = min(pp + chunk, k_off[k + 1])
                rows.append((k, pp, qq)); tag.append(-1)
                pp = qq
    rows = np.array(rows, dtype=np.int64); tag = np.array(tag)
    n_items = len(rows)
    ids_by_block = [np.nonzero(tag == b)[0] for b in range(nb)]
    sparse_ids = list(np.nonzero(tag < 0)[0])
    G = len(dense)
    order = []
    # launch slots: 8 XCDs round-robin; block b goes to XCD b % 8: emit rounds of 8 blocks, interleaved so that slot 8 j + x is
    # item j of the round's x-th block; sparse items fill the slots of blocks that are missing in the last round
    for s0 in range(0, nb, 8):
        blocks = list(range(s0, min(s0 + 8, nb)))
        for j in range(G):
            for xc in range(8):
                if xc < len(blocks):
                    order.append(ids_by_block[blocks[xc]][j])
                elif sparse_ids:
                    order.append(sparse_ids.pop())
    order += sparse_ids
    order = np.array(order, dtype=np.int64)
    assert sorted(order.tolist()) == list(range(n_items))
    items_np = np.stack([rows[:, 0], rows[:, 1], rows[:, 2], order]).astype(np.int32)
    items = torch.from_numpy(items_np).cuda()
    io = np.zeros(m.K + 1, dtype=np.int32)
    for k in range(m.K):
        io[k + 1] = io[k] + int((rows[:, 0] == k).sum())
    item_off = torch.from_numpy(io).cuda()
    print(f"hybrid items: {len(dense)} dense offsets x {nb} blocks of {HYB} rows + {int((tag < 0).sum())} sparse items = {n_items}; "
          f"dense item pairs min/mean/max {int((rows[tag>=0,2]-rows[tag>=0,1]).min())}/{int((rows[tag>=0,2]-rows[tag>=0,1]).mean())}/{int((rows[tag>=0,2]-rows[tag>=0,1]).max())}")
slabs = _lib.load().lidog_sconv_wgrad_slabs(Cin, Cout, n_items)
partial = torch.empty((max(slabs, 1), Cin, Cout), device="cuda"); gW = torch.empty((m.K, Cin, Cout), device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(6):
    if it == 1: ev[0].record()
    call("lidog_sconv_wgrad", ptr(x), ptr(m.pair_in), ptr(g), ptr(m.pair_out), ptr(items), n_items, ptr(item_off), m.K,
         Cin, Cout, ptr(partial), ptr(gW))
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 5
print(f"s{s} {Cin}->{Cout}: P {m.P} items {n_items} slabs {slabs}: {ms:.3f} ms, {2e-9 * m.P * Cin * Cout / ms:.1f} TF/s "
      f"(gathers if nothing hits: {4e-6 * m.P * (Cin + Cout):.0f} MB, distinct rows {4e-6 * (m.n_in * Cin + m.n_out * Cout):.0f} MB, "
      f"partials {4e-6 * slabs * Cin * Cout:.0f} MB)")
