"""Round 4: how many run tails per wave does the aggregation pass of the hash-grid backward see at every level?  Synthetic PSF clouds,
samples sorted by the Morton code of their finest-level cell, runs cut at 16-lane rows as in the kernel (hashgrid_bwd_aggregate::prepare).
Prints per level: lattice box volume, distinct vertices, tails per wave (rows of 16 / whole wave).  CPU only: python tools/sim_tails_per_wave.py"""
import numpy as np
rng = np.random.default_rng(0)
NW=200; base, scale_f, L = 9, 1.26, 16
scales = [np.float32(np.exp2(l * np.log2(scale_f)) * base - 1) for l in range(L)]
def spread3(x):
    x = x & 0xff
    x = (x ^ (x << 8)) & 0x0300f00f
    x = (x ^ (x << 4)) & 0x030c30c3
    x = (x ^ (x << 2)) & 0x09249249
    return x
T=np.zeros((L,NW*4)); V=np.zeros((L,NW)); box=np.zeros((L,NW)); TW=np.zeros((L,NW*4))
for w in range(NW):
    c = rng.random(3) * 110 + 10
    pts = (c + rng.standard_normal((256, 3)) * np.array([0.77, 0.77, 1.27])) / 130.0
    pts = np.clip(pts, 0, 1).astype(np.float32)
    cf = np.floor(pts * scales[L - 1] + np.float32(0.5)).astype(np.int64)
    code = spread3(cf[:, 0]) | (spread3(cf[:, 1]) << 1) | (spread3(cf[:, 2]) << 2)
    order = np.argsort(code * 256 + np.arange(256), kind="stable")
    pts = pts[order]
    for l in range(L):
        cell = np.floor(pts * scales[l] + np.float32(0.5)).astype(np.int64)
        ex = cell.max(0) - cell.min(0)
        box[l,w] = np.prod(ex+2)
        verts=set()
        for k in range(8):
            for r in cell + np.array([k&1,(k>>1)&1,k>>2]): verts.add(tuple(r))
        V[l,w]=len(verts)
        for wave in range(4):
            t=0; tw=0
            cw = cell[64*wave:64*wave+64]
            chg = np.ones(64,bool); chg[:-1] = (cw[1:]!=cw[:-1]).any(1)
            tw = chg.sum()  # tails if runs spanned the whole wave
            for row in range(4):
                cc = cw[16*row:16*row+16]
                tail = np.ones(16,bool); tail[:-1] = (cc[1:]!=cc[:-1]).any(1)
                t += tail.sum()
            T[l,w*4+wave]=t; TW[l,w*4+wave]=tw
for l in range(L):
    print(f"level {l:2d} res {scales[l]+1:6.1f}: box vol mean {box[l].mean():7.0f} max {box[l].max():6.0f} | verts {V[l].mean():6.0f} | tails/wave (rows) mean {T[l].mean():5.1f} p90 {np.percentile(T[l],90):4.0f} max {T[l].max():3.0f} | wave-wide runs {TW[l].mean():5.1f}")
