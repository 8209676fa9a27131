"""LDS bank conflicts of the radial sweep's gathers (wave_ops.h wave_monotonic, compact one-trip levels), simulated:
ds_read_b32 = two groups of 32 lanes over 32 banks, identical addresses broadcast; random centres, levels 1..40.
    python tools/sweep_banks.py [tile row stride in floats]"""
import numpy as np
import sys
H=W=64; LW=int(sys.argv[1]) if len(sys.argv) > 1 else 66
def walker_at(xmajor, mi, cy, cx, half, enabled=True):
    cmin = cy if xmajor else cx; cmaj = cx if xmajor else cy
    dimmin = H if xmajor else W; dimmaj = W if xmajor else H
    strmin = LW if xmajor else 1; strmaj = 1 if xmajor else LW
    d = -1 if half else 1
    Yb = mi - cmin
    s = -1 if Yb > 0 else 1
    b = abs(Yb)
    valid = enabled and 0 <= mi < dimmin
    ok1 = 0 <= mi + s < dimmin; ok3 = 0 <= mi - s < dimmin
    amin = max(b,1) if xmajor else b+1
    lim = cmaj if half else dimmaj-1-cmaj
    base = mi*strmin + cmaj*strmaj
    step = d*strmaj
    return dict(step=step, sa=-step, off1=(-step + s*strmin) if ok1 else 0, off3=(-step - s*strmin) if ok3 else 0,
                off4=(s*strmin if b>0 else 0), famin=(amin if valid else 1e9), flim=lim, base=base, b=b)
def compact(e, cy, cx, lane):
    wedge = lane>>4; side=(lane>>3)&1; j=lane&7
    xmajor = wedge<2
    b = 2*j+e
    mi = (cy if xmajor else cx) + (-b if side else b)
    return walker_at(xmajor, mi, cy, cx, wedge&1, not(side and b==0))
def conflicts(addrs, act=None):
    # ds_read_b32: two groups of 32 lanes, 32 banks; identical addresses broadcast
    tot=0
    for g in (range(0,32), range(32,64)):
        banks={}
        for l in g:
            if act is not None and not act[l]: continue
            a=addrs[l]
            banks.setdefault(a%32,set()).add(a)
        tot += max([len(v) for v in banks.values()] or [1])
    return tot   # cycles (2 = conflict-free)
rng=np.random.RandomState(0)
res_all=[];res_act=[]
for trial in range(200):
    cy,cx = rng.randint(8,56,2)
    w=[[compact(e,cy,cx,l) for l in range(64)] for e in (0,1)]
    for level in range(1,41):
        e = level&1
        ws=w[e]
        addr={k:[] for k in "p p1 p2 p3 p4".split()}; act=[]
        for l in range(64):
            wk=ws[l]; a=(level-wk['b'])>>1
            p=wk['base']+a*wk['step']
            act.append(wk['famin']<=a<=wk['flim'])
            addr['p'].append(p); addr['p2'].append(p+wk['sa']); addr['p1'].append(p+wk['off1']); addr['p3'].append(p+wk['off3']); addr['p4'].append(p+wk['off4'])
        for k in addr:
            aa=[x & 0xffffffff for x in addr[k]]
            res_all.append(conflicts(aa)); res_act.append(conflicts(aa,act))
print("cycles per ds_read_b32 (2 = conflict-free): all lanes issue: mean %.2f  p90 %d;  active lanes only: mean %.2f p90 %d" % (np.mean(res_all), np.percentile(res_all,90), np.mean(res_act), np.percentile(res_act,90)))
