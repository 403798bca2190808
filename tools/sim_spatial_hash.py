"""How many vertices of a PSF cloud miss their first slot in the hashed merge table of the hash-grid backward's aggregation
pass (csrc/hashgrid.hip, NESVOR_HG_SPATIAL), CPU only: 200 synthetic clouds of 256 samples (sigma 0.77 / 0.77 / 1.27 mm in a
130 mm cube, the distribution of tools/hg_variants.py), levels 12-15 of the headline grid (base 9, scale 1.26), 1024 slots.

  multiplicative:  slot = (entry index * 2654435761) >> 22          (rounds 1-4)
  spatial:         slot = x mod 2^a | (y mod 2^b) << a | (z mod 2^c) << (a + b), bits dealt by the box extent (round 5)

    python tools/sim_spatial_hash.py
"""
import numpy as np

rng = np.random.default_rng(0)
n_clouds, slots_log2 = 200, 10
centre = rng.random((n_clouds, 1, 3)) * 110 + 10
u = ((centre + rng.standard_normal((n_clouds, 256, 3)) * np.array([0.77, 0.77, 1.27])) / 130.0).clip(0, 1)
base, scale, T = 9, 1.26, 1 << 19
corner = np.array([[k & 1, (k >> 1) & 1, k >> 2] for k in range(8)])


def window_bits(ext):
    bits = [0, 0, 0]
    for _ in range(slots_log2):
        r = [(int(e) << 12) >> b for e, b in zip(ext, bits)]
        d = 2 if r[2] >= r[0] and r[2] >= r[1] else (1 if r[1] >= r[0] else 0)
        bits[d] += 1
    return bits


for level in (12, 13, 14, 15):
    g = np.floor(u * (base * scale**level - 1) + 0.5).astype(np.int64)
    n_vert = miss_mult = miss_spatial = 0
    for cells in g:
        v = np.unique((cells[:, None, :] + corner[None]).reshape(-1, 3), axis=0)
        n_vert += len(v)
        idx = np.unique((v[:, 0] ^ (v[:, 1] * 2654435761) ^ (v[:, 2] * 805459861)) % T)
        _, cnt = np.unique(((idx * 2654435761) & 0xFFFFFFFF) >> (32 - slots_log2), return_counts=True)
        miss_mult += int((cnt - 1).sum())
        a, b, c = window_bits(v.max(0) - v.min(0) + 1)
        slot = (v[:, 0] % (1 << a)) | ((v[:, 1] % (1 << b)) << a) | ((v[:, 2] % (1 << c)) << (a + b))
        _, cnt = np.unique(slot, return_counts=True)
        miss_spatial += int((cnt - 1).sum())
    print(f"level {level}: {n_vert / n_clouds:6.0f} vertices per cloud; first-slot misses per cloud: multiplicative {miss_mult / n_clouds:6.1f}, "
          f"spatial {miss_spatial / n_clouds:6.1f}")
