"""launch-level diagnostics of k_fit2x (SCARLET_STAMPS=1): workgroup lifetimes, mean residency per CU, start-time spread"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from scarlet_amd import synth
from scarlet_amd.batch import BlendBatch
S, iters = int(os.environ.get("STAMP_S", "10000")), int(os.environ.get("STAMP_ITERS", "50"))
d = synth.make_batch(0, 512)
reps = (S + 511) // 512
imgs = np.tile(d["images"], (reps, 1, 1, 1))[:S]; cen = np.tile(d["centers"], (reps, 1, 1))[:S]
b = BlendBatch(imgs, cen)
b.init_extended(np.ones(5) * .1)
b.fit(10, e_rel=0, check_every=0)
torch.cuda.synchronize()
b.workspace[:S * 16 * 8].zero_()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); b.fit(iters, e_rel=0, check_every=0); t1.record()
torch.cuda.synchronize()
st = b.workspace[:S * 16 * 8].view(torch.int64).view(S, 16).cpu().numpy()
start, last_start, end = st[:, 14].astype(float), st[:, 7].astype(float), st[:, 11].astype(float)      # 100 MHz ticks
life = (end - start) / 100.0                    # us
wall = (end.max() - start.min()) / 100.0
hw = st[:, 15]
hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 0x1) << 4) | (((hwid >> 13) & 0x7) << 5)     # cu_id, sh_id, se_id
key = xcc * 1024 + cu
print("event-timed launch %.3f ms; in-kernel first start -> last end %.3f ms" % (t0.elapsed_time(t1), wall / 1e3))
print("workgroup lifetime: mean %.1f us (%.2f us per iteration), p10 %.1f, p90 %.1f" % (life.mean(), life.mean() / iters, np.percentile(life, 10), np.percentile(life, 90)))
print("last iteration alone (stamps 0 -> 6, realtime): mean %.2f us" % ((end - last_start).mean() / 100.0))
print("sum of lifetimes / (wall x 256 CUs) = mean resident workgroups per CU: %.3f" % (life.sum() / (wall * 256)))
uniq = np.unique(key)
print("distinct (xcc, se, sh, cu) seen: %d; workgroups per CU: min %d max %d" % (len(uniq), min((key == u).sum() for u in uniq), max((key == u).sum() for u in uniq)))
per_xcc = [int((xcc == x).sum()) for x in range(8)]
print("workgroups per XCC:", per_xcc)
order = np.argsort(start)
ss = (start[order] - start.min()) / 100.0
print("start times (us) of workgroups #0, 256, 511, 512, 600, 1000, 5000, last: ", [round(float(ss[i]), 1) for i in (0, 256, 511, 512, 600, 1000, 5000, S - 1)])
# per CU: busy fraction with 2 resident
for x in range(2):
    sel = key == uniq[x]
    ev = sorted([(t, 1) for t in start[sel]] + [(t, -1) for t in end[sel]])
    cur, prev, acc = 0, ev[0][0], {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
    for t, dlt in ev:
        acc[min(cur, 3)] += t - prev; prev = t; cur += dlt
    tot = sum(acc.values())
    print("CU %d: time with 0/1/2/3+ workgroups resident: %s" % (uniq[x], {k: round(v / tot, 3) for k, v in acc.items()}))
# refill gaps: how long a CU stays at one resident workgroup after a workgroup ends (excluding the final tail)
gaps, tails = [], []
for u in uniq:
    sel = key == u
    ev = sorted([(t, 1) for t in start[sel]] + [(t, -1) for t in end[sel]])
    cur, drop = 0, None
    last_start_t = max(start[sel])
    for t, dlt in ev:
        cur += dlt
        if dlt < 0 and cur == 1:
            drop = t
        if dlt > 0 and cur == 2 and drop is not None:
            gaps.append((t - drop) / 100.0); drop = None
    tails.append((max(end[sel]) - last_start_t) / 100.0)
gaps = np.array(gaps)
print("refill gaps (us) between a workgroup's end and the next start on the same CU: n %d, mean %.1f, median %.1f, p90 %.1f, max %.1f; sum per CU %.1f us" % (
    len(gaps), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max(), gaps.sum() / 256))
print("per-CU end of last workgroup (us after launch start): min %.0f  median %.0f  max %.0f" % tuple(
    np.percentile([(max(end[key == u]) - start.min()) / 100.0 for u in uniq], [0, 50, 100])))
xe = [(max(end[xcc == x]) - start.min()) / 100.0 for x in range(8)]
print("per-XCC end (us):", [round(v) for v in xe])
xs = [(np.sort(start[xcc == x])[-1] - start.min()) / 100.0 for x in range(8)]
print("per-XCC start of its last workgroup (us):", [round(v) for v in xs])
if os.environ.get("STAMP_PERIOD"):
    per = b.workspace[S * 16 * 8:S * 16 * 8 + S * 2 * 8].view(torch.int64).view(S, 2).cpu().numpy().astype(float)
    dt = np.abs(per[:, 0] - per[:, 1]) / 100.0
    print("start-to-start period of the last two iterations: mean %.2f us, median %.2f; body of the last (stamps 0 -> 6): %.2f us" % (
        dt.mean(), np.median(dt), (end - last_start).mean() / 100.0))
