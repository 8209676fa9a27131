import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth
from scarlet_amd.batch import BlendBatch
S = int(os.environ.get("STAMP_S", "10000"))
d = synth.make_batch(0, 512)
reps = (S + 511) // 512
imgs = np.tile(d["images"], (reps, 1, 1, 1))[:S]; cen = np.tile(d["centers"], (reps, 1, 1))[:S]
kw = {}
if len(sys.argv) > 1 and sys.argv[1] == "nocons": kw = dict(symmetric=False, monotonic=False)
b = BlendBatch(imgs, cen, **kw)
b.init_extended(np.ones(5) * .1)
b.fit(int(os.environ.get("STAMP_PRE", "3")), e_rel=0, check_every=0)
torch.cuda.synchronize()
b.workspace[:S * 16 * 8].zero_()
# STAMP_ITERS > 1: the multi-iteration kernel (k_fit2x); the stamps are those of the launch's LAST iteration
b.fit(int(os.environ.get("STAMP_ITERS", "1")), e_rel=0, check_every=0)
torch.cuda.synchronize()
st = b.workspace[:S * 16 * 8].view(torch.int64).view(S, 16).cpu().numpy()
if int(os.environ.get("STAMP_ITERS", "1")) > 1 and S > 2048:
    # multi-iteration launch: the last iteration of the scenes that finish while the chip is still full (the last
    # ~1000 scenes of the queue finish beside emptying CUs and run faster)
    st = st[:S - 1024]
dt = np.diff(st[:, :7], axis=1)
if st[:, 10].any():
    print("P2 wave0: pre-sym %d  sym %d  sweep %d  tail %d" % ((st[:, 8] - st[:, 4]).mean(), (st[:, 9] - st[:, 8]).mean(),
          (st[:, 10] - st[:, 9]).mean(), (st[:, 5] - st[:, 10]).mean()))
if st[:, 12].any():
    ok = st[:, 12] > 0
    if st[:, 11].any():
        print("sym wave0: vectors %d  rank1 %d  gemm1 %d  gemm2 %d  (n=%d)" % ((st[ok, 12] - st[ok, 8]).mean(), (st[ok, 13] - st[ok, 12]).mean(),
              (st[ok, 14] - st[ok, 13]).mean(), (st[ok, 9] - st[ok, 14]).mean(), ok.sum()))
    else:
        print("sym (pair kernel): to B1 %d  B1->B2 (rank1 z + gemm1) %d  B2->B3 (gemm2 + epilogue) %d" % ((st[ok, 12] - st[ok, 8]).mean(),
              (st[ok, 13] - st[ok, 12]).mean(), (st[ok, 9] - st[ok, 13]).mean()))
        o2 = st[:, 15] > 0
        print("   wave0: z %d  gemm1 %d  wait-B2 %d  (n=%d)" % ((st[o2, 14] - st[o2, 12]).mean(), (st[o2, 15] - st[o2, 14]).mean(),
              (st[o2, 13] - st[o2, 15]).mean(), o2.sum()))
    if st[:, 11].any(): print("P2 wave0 own tail:", (st[:, 11] - st[:, 10]).mean().round(0))
print("phase cycles mean:", dt.mean(axis=0).round(0), " total", (st[:, 6] - st[:, 0]).mean())
print("phase cycles p90 :", np.percentile(dt, 90, axis=0).round(0))
if st[:, 7].any() and st[:, 11].any():
    cyc, ticks = (st[:, 6] - st[:, 0]).astype(float), (st[:, 11] - st[:, 7]).astype(float)
    ok = ticks > 0
    print("in-kernel clock (shader cycles per 100 MHz tick x 100 MHz): median %.0f MHz, p10 %.0f, p90 %.0f" % (
        np.median(cyc[ok] / ticks[ok]) * 100, np.percentile(cyc[ok] / ticks[ok], 10) * 100, np.percentile(cyc[ok] / ticks[ok], 90) * 100))
