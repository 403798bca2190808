"""Round 4: simulate the LDS bank behaviour of the merge-table inserts of hashgrid_bwd_aggregate (box rounds) on synthetic PSF clouds:
Morton-sorted samples, run tails per 16-lane row, box slot addressing, 64-bit LDS accesses served in 4 x 16 contiguous lanes on 32
four-byte banks (MI355X_MICROARCH.md, LDS).  Prints cycles per lane group relative to conflict-free for the kernel's layout and for
three alternatives (x-extent padded to odd, pad-to-4 with a fixed xy pitch residue, XOR swizzle).  CPU only:  python tools/sim_lds_conflicts.py"""
import numpy as np, sys
rng = np.random.default_rng(0)
NW = 300  # clouds
base, scale_f, L = 9, 1.26, 16
scales = [np.float32(np.exp2(l * np.log2(scale_f)) * base - 1) for l in range(L)]
def spread3(x):
    x = x & 0xff
    x = (x ^ (x << 8)) & 0x0300f00f
    x = (x ^ (x << 4)) & 0x030c30c3
    x = (x ^ (x << 2)) & 0x09249249
    return x
def group_cycles(addr_bytes_list, width):
    """cycles of one lane group: max over banks of distinct addresses; bank = (a/4) mod 32; width bytes per lane"""
    banks = {}
    for a in addr_bytes_list:
        for w in range(width // 4):
            b = ((a // 4) + w) % 32
            banks.setdefault(b, set()).add(a)
    return max((len(s) for s in banks.values()), default=0)
tot = {}
for variant in ("base", "nx_odd", "pad4_16", "xor"):
    tot[variant] = np.zeros((L, 3))  # cycles, ideal (groups with any active), n
for w in range(NW):
    c = rng.random(3) * 110 + 10
    pts = (c + rng.standard_normal((256, 3)) * np.array([0.77, 0.77, 1.27])) / 130.0
    pts = np.clip(pts, 0, 1).astype(np.float32)
    posf = pts * scales[L - 1] + np.float32(0.5)
    cf = np.floor(posf).astype(np.int64)
    code = spread3(cf[:, 0]) | (spread3(cf[:, 1]) << 1) | (spread3(cf[:, 2]) << 2)
    order = np.argsort(code * 256 + np.arange(256), kind="stable")
    pts = pts[order]
    for l in range(L):
        pos = pts * scales[l] + np.float32(0.5)
        cell = np.floor(pos).astype(np.int64)
        lo = cell.min(0); ex = cell.max(0) - lo
        nx, ny, nz = ex + 2
        if nx * ny * nz > 1024:
            continue
        for variant in tot:
            nx_, ny_ = nx, ny
            if variant == "nx_odd":
                nx_ = nx | 1
            if variant == "pad4_16":
                nx_ = (nx + 3) // 4 * 4
                # nxy = 16 mod 32 -> choose ny_ >= ny with nx_*ny_ % 32 == 16 if possible
                ny_ = ny
                while (nx_ * ny_) % 32 != 16 and ny_ < ny + 8:
                    ny_ += 1
            nxy_ = nx_ * ny_
            if nxy_ * nz > 1024 and variant != "base":
                nx_, ny_, nxy_ = nx, ny, nx * ny
            rel = cell - lo
            s0 = (rel[:, 2] * ny_ + rel[:, 1]) * nx_ + rel[:, 0]
            # run tails within 16-lane rows
            cyc = ideal = 0
            for row in range(16):
                r = slice(16 * row, 16 * row + 16)
                cc = cell[r]
                tail = np.ones(16, bool)
                tail[:-1] = (cc[1:] != cc[:-1]).any(1)
                for k in range(8):
                    s = s0[r][tail] + (k & 1) + ((k >> 1) & 1) * nx_ + (k >> 2) * nxy_
                    if variant == "xor":
                        s = s ^ ((s >> 4) & 15)
                    g = group_cycles(list(8 * s), 8)
                    cyc += g; ideal += 1
            tot[variant][l] += (cyc, ideal, 1)
for variant, t in tot.items():
    print(variant)
    for l in range(L):
        if t[l, 2]:
            print(f"  level {l:2d}: clouds {int(t[l,2]):4d}  cycles/ideal {t[l,0]/t[l,1]:.2f}")
    print(f"  total cycles {t[:,0].sum():.0f}  ideal {t[:,1].sum():.0f}  ratio {t[:,0].sum()/t[:,1].sum():.3f}")
