import sys, numpy as np
sys.path.insert(0, '/root/repo')
from lidog_amd import synth
from lidog_amd.bev import pixel_luts
b = synth.make_batch([0], "kitti120k", device="cpu")
c = b["coords_int"].numpy()
lx, ly, lo, H = pixel_luts(50.0, 0.05)
px = lx[c[:,1]-lo]; py = ly[c[:,2]-lo]
ok = (px>=0)&(py>=0)
px, py = px[ok], py[ok]
pix = np.unique(py.astype(np.int64)*H+px)
print("voxels", len(c), "occupied pixels", len(pix))
W=H
f = (pix[:,None]*96 + np.arange(96)[None,:]).ravel()
cp = f//(H*W); r = f%(H*W); yp = r//W; xp = r%W
occ = np.zeros((96,H,W), bool); occ[cp,yp,xp]=True
# maxpool 5 s3 p1 occupancy
Ho = (H+2-5)//3+1
pad = np.zeros((96,H+2,W+2),bool); pad[:,1:-1,1:-1]=occ
pooled = np.zeros((96,Ho,Ho),bool)
for dy in range(5):
    for dx in range(5):
        pooled |= pad[:, dy:dy+3*Ho:3, dx:dx+3*Ho:3][:, :Ho, :Ho]
print("pooled occupancy", pooled.mean())
# conv1: 3x3 s2 p1 -> 333; output tile of 128 consecutive (yo*333+xo) pixels; channel needed if any input cell in receptive field nonzero
Hc=(Ho+2-3)//2+1
pp = np.zeros((96,Ho+2,Ho+2),bool); pp[:,1:-1,1:-1]=pooled
need = np.zeros((96,Hc,Hc),bool)
for dy in range(3):
    for dx in range(3):
        need |= pp[:, dy:dy+2*Hc:2, dx:dx+2*Hc:2][:, :Hc, :Hc]
print("output pixel x channel needs:", need.mean())
flat = need.reshape(96,-1)
for TM in (32,64,128,256):
    n = flat.shape[1]//TM*TM
    t = flat[:,:n].reshape(96,-1,TM).any(2)
    print("M tile", TM, "fraction of (tile,channel) non-zero:", t.mean(), " tiles fully empty:", (~t.any(0)).mean())
# 2D tiles 8x16
for th,tw in ((8,16),(16,8),(4,32),(8,8)):
    hh=Hc//th*th; ww=Hc//tw*tw
    t = need[:,:hh,:ww].reshape(96,hh//th,th,ww//tw,tw).any(4).any(2)
    print("tile %dx%d"%(th,tw), t.mean())
# groups of 3 channels
t = flat[:, :flat.shape[1]//128*128].reshape(32,3,-1,128).any(3).any(1)
print("M128 x 3-channel groups nonzero:", t.mean())
print("---- wgrad grouping")
n = flat.shape[1]//128*128
t128 = flat[:, :n].reshape(96,-1,128).any(2)   # [96, tiles]
for GC in (1,2,3,4,6,7,8,12,14,16,24,32):
    ng = (96+GC-1)//GC
    pad = np.zeros((ng*GC, t128.shape[1]), bool); pad[:96]=t128
    a = pad.reshape(ng, GC, -1).any(1)
    frac = a.mean()
    ncols = -(-9*GC//32)*32
    print("GC %2d groups %2d active %.3f  A-traffic %.2f (dense 6.75)  MFMA units %.0f (dense 864)" % (GC, ng, frac, ng*frac, ng*frac*ncols))
print("---- structural support of the FIRST convolution's OUTPUT (VERDICT r4 item 2: is there a constant background behind")
print("     Conv2d(bias=False) -> BatchNorm2d -> ReLU that the second convolution could skip?)")
s1 = need.any(0)   # an output pixel of conv1 is exactly zero before BatchNorm only if NO channel has a cell in its 3x3 window
print("conv1 output pixels with a non-empty window in at least one of the 96 channel planes: %.4f of %d x %d" % (s1.mean(), Hc, Hc))
print("active channels per conv1 output pixel: mean %.1f, min %d" % (need.sum(0).mean(), need.sum(0).min()))
H2 = (Hc + 2 - 3) // 2 + 1
p2 = np.zeros((Hc + 2, Hc + 2), bool); p2[1:-1, 1:-1] = s1
s2 = np.zeros((H2, H2), bool)
for dy in range(3):
    for dx in range(3):
        s2 |= p2[dy:dy + 2 * H2:2, dx:dx + 2 * H2:2][:H2, :H2]
print("conv2 output pixels whose 3x3 window touches that support: %.4f of %d x %d" % (s2.mean(), H2, H2))
print("(the `.view(1, C, H, W)` scramble of sparse2super, minkunet_bev.py:221, spreads every occupied BEV pixel over 96 cells of")
print(" ONE channel plane at a plane-dependent place: each plane is ~5 %% occupied, their union covers the image)")
