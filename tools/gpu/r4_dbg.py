import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0]); sys.path.insert(0, __file__.rsplit("/tools/", 1)[0] + "/tests")
import torch
from conftest import DENSE_INT, DENSE_TINYINT, GT, RawColumn, blocks_of
from immutable3_amd import native, synth
n = 13_100 * 1024 - 333
a = synth.uniform_int30(31, n)
c = synth.uniform_below(33, n, 100, np.int8)
br = blocks_of(n, 1024)
ctx = native.Context(0)
seg = native.DeviceSegment(ctx, [RawColumn(DENSE_INT, 4, a, br).native(), RawColumn(DENSE_TINYINT, 1, c, br).native()])
keep = (c > 29) & (a > 0.2 * 2 ** 30)
rows = np.flatnonzero(keep)
want = np.packbits(keep, bitorder="little")
for P in (2, 3, 6):
    ctx.set_tuning(200 + P, 0)
    q = native.DeviceQuery(ctx, seg, [1, 0], [(0, GT, 29.0), (1, GT, float(0.2 * 2 ** 30))], [1, 0], 0)
    ctx.set_tuning(0, 0)
    bad = 0
    for it in range(40):
        q.run()
        cnt = q.count()
        bm = q.bitmap().view(np.uint8)
        w = want[: bm.size]
        if cnt != rows.size or not (bm[: w.size] == w).all():
            bad += 1
            diff = np.flatnonzero(bm[: w.size] != w)
            tiles = np.unique(diff // 128)
            print(f"P {P} it {it}: count {cnt} want {rows.size} (diff {rows.size - cnt}); bitmap bytes differ {diff.size}; tiles {tiles[:20]} n_tiles {tiles.size} plan {q.plan()}", flush=True)
            if tiles.size:
                t = int(tiles[0]); span = t // (8 * P); wave = (t // P) % 8
                print(f"   first tile {t}: span {span} wg {span % 256} round {span // 256} wave {wave} j {t % P}; got bytes {bm[t*128:t*128+8]} want {w[t*128:t*128+8]}", flush=True)
    print(f"P {P}: {bad} bad of 40", flush=True)
    q.close()
