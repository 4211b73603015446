"""CPU model of K1's part scheme (scan_segments.h, P3): for a picture's quantized blocks (oracle), how many parts a
segment makes under a merging rule, how many rounds of 256 they take, and how many wave-iterations the sorted,
boustrophedon deal costs (sum over rounds and waves of the longest walk in the wave) -- the quantities the walk's
instruction count is made of.  Needs no GPU.
  python tools/part_model.py [struct|noise] [w h] [mode 1|3|4] [q]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc, synth

kind = sys.argv[1] if len(sys.argv) > 1 else "struct"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
q = float(sys.argv[5]) if len(sys.argv) > 5 else 75.0
o = orc.oracle()
img = (synth.g_struct if kind == "struct" else synth.g_noise)(w, h)
cache = "/tmp/model/zz_%s_%d_%d_%d_%g.npy" % (kind, w, h, mode, q)
if os.path.exists(cache):
    zz = np.load(cache)
else:
    zz = o.scan_coeffs(img, o.quality_matrices(q), yuv_mode=mode)
    os.makedirs("/tmp/model", exist_ok=True); np.save(cache, zz)
bpm = {1: 6, 3: 3, 4: 1}[mode]
seg_blocks = 246
nz = zz != 0
nz[:, 0] = False
cnt_q = nz.reshape(-1, 4, 16).sum(2)                      # symbols per quarter
pos = np.arange(64)
first_q = np.where(nz.reshape(-1, 4, 16), np.arange(16)[None, None, :], 99).min(2)      # local first position (99 = none)
last_q = np.where(nz.reshape(-1, 4, 16), np.arange(16)[None, None, :] + 1, 0).max(2)    # local end (0 = none)
nb = len(zz)
print("%s %dx%d mode %d q%g: %d blocks, %.2f symbols per block (max %d), blocks with upper half non-zero %.1f %%" % (
    kind, w, h, mode, q, nb, nz.sum() / nb, nz.sum(1).max(), 100.0 * (cnt_q[:, 2:].sum(1) > 0).mean()))

def parts_of(rule):
    """list of arrays: per block the symbol counts of its parts"""
    c = cnt_q
    out = []
    end0 = np.maximum(last_q[:, 0], 1)
    if rule == "quarters":
        p = [c[:, 0], c[:, 1], c[:, 2], c[:, 3]]
        valid = [np.ones(nb, bool), c[:, 1] > 0, c[:, 2] > 0, c[:, 3] > 0]
    elif rule.startswith("halves"):
        cap = int(rule[6:] or 16)
        m01 = (c[:, 1] > 0) & (first_q[:, 1] < end0) & (c[:, 0] + c[:, 1] <= cap)
        m23 = (c[:, 3] > 0) & (first_q[:, 3] < last_q[:, 2]) & (c[:, 2] + c[:, 3] <= cap)
        p = [np.where(m01, c[:, 0] + c[:, 1], c[:, 0]), np.where(m01, 0, c[:, 1]), np.where(m23, c[:, 2] + c[:, 3], c[:, 2]), np.where(m23, 0, c[:, 3])]
        valid = [np.ones(nb, bool), p[1] > 0, p[2] > 0, p[3] > 0]
    elif rule.startswith("greedy"):
        # consecutive quarters joined while the sum stays <= cap (ignores the ZRL condition: an upper bound of what merging can give)
        cap = int(rule[6:].rstrip("p") or 16)
        p = [np.zeros(nb, int) for _ in range(4)]; valid = [np.zeros(nb, bool) for _ in range(4)]
        cur = c[:, 0].copy(); curq = np.zeros(nb, int); valid[0][:] = True
        pp = np.zeros((nb, 4), int); vv = np.zeros((nb, 4), bool); vv[:, 0] = True; pp[:, 0] = c[:, 0]
        owner = np.zeros(nb, int)
        span = 2 if rule.endswith("p") else 4                  # "p": a part spans at most two quarters (a 32-bit mask)
        for qq in (1, 2, 3):
            join = (c[:, qq] > 0) & (pp[np.arange(nb), owner] + c[:, qq] <= cap) & (qq - owner < span)
            new = (c[:, qq] > 0) & ~join
            pp[np.arange(nb)[join], owner[join]] += c[join, qq]
            pp[np.arange(nb)[new], qq] = c[new, qq]; vv[np.arange(nb)[new], qq] = True; owner[new] = qq
        p = [pp[:, i] for i in range(4)]; valid = [vv[:, i] for i in range(4)]
    return np.stack(p, 1), np.stack(valid, 1)

def model(rule, per_part=110, per_sym=27, coarse=False):
    pcs, valid = parts_of(rule)
    nseg = (nb + seg_blocks - 1) // seg_blocks
    tot_units = tot_rounds = tot_iter = tot_roundslots = 0
    for s in range(nseg):
        sl = slice(s * seg_blocks, min(nb, (s + 1) * seg_blocks))
        u = pcs[sl][valid[sl]]
        if coarse:      # bins of the sort: one per count up to 16, then 17-20, 21-24, 25-32 (stream order inside a bin)
            key = np.where(u <= 16, u, 17 + np.minimum((u - 17) >> 2, 2))
            u = u[np.argsort(-key, kind="stable")]
        else:
            u = np.sort(u)[::-1]
        n = len(u); tot_units += n
        rounds = (n + 255) // 256; tot_rounds += rounds
        for g in range((n + 63) // 64):
            tot_iter += u[g * 64:(g + 1) * 64].max()
        tot_roundslots += (n + 63) // 64
    syms = nz.sum()
    print("%-10s parts/block %.2f  rounds/segment %.2f  wave-groups/segment %.2f  wave-iterations/segment %.1f (perfect %.1f)  model VALU/wave %.0f" % (
        rule, tot_units / nb, tot_rounds / nseg, tot_roundslots / nseg, tot_iter / nseg, syms / nseg / 64,
        (tot_iter * per_sym + tot_roundslots * per_part) / nseg / 4))

for r in ("quarters", "halves16", "halves24", "halves32"):
    model(r)
for r in ("halves20", "halves24", "halves32"):
    print("coarse bins:", end=" "); model(r, coarse=True)
