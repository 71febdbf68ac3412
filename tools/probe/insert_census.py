"""CPU census for VERDICT r3 item 2 (no GPU; ~4 min, ~10 GB): batch `nb` of the config-2 stream on the table the first nb
batches built.
 (b) ticket-free inserts: a `used` ticket is needed only when a row COULD cross the reference's threshold (src/smatrix.c:346)
     in this batch.  Per row: room = size/2+1 - used at the start of the batch; which share of the batch's inserts lands in
     rows whose room covers (i) the row's true number of new keys (what an oracle would know), (ii) the row's op count in
     the batch (a bound a per-batch row histogram would give), (iii) a fixed multiple of the row's insert count of the
     PREVIOUS batch (an estimate, not a bound)?
 (c) hot/cold split of the LDS fold: which share of the (tile, key) entries has exactly one op in its tile?
 (a) the growth round: deferred ops and growing rows by table size.
python tools/probe/insert_census.py [nb]"""
import numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from libsmatrix_amd import Stream
B = 1 << 24
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 15
gen = Stream("zipf", 12345, 1000000, 1.1, 1)
t0 = time.time()
def keys_of(b):
    x, y = gen.fill(b * B, B)
    return (x.astype(np.uint64) << np.uint64(32)) | y.astype(np.uint64)
allk = np.zeros(0, np.uint64)
prev_new = None
for b in range(nb):
    k = np.unique(keys_of(b))
    new = np.setdiff1d(k, allk, assume_unique=True)
    if b == nb - 1:
        prev_new = new
    allk = np.union1d(allk, new)
    print('batch', b, 'nnz', allk.size, '%.0fs' % (time.time() - t0), flush=True)
rows_before = (allk >> np.uint64(32)).astype(np.uint32)
ur, used = np.unique(rows_before, return_counts=True)
def lg_of(n):
    lg = np.full(n.shape, 4, np.int64)
    while True:
        m = n > (1 << lg) // 2 + 1
        if not m.any(): break
        lg[m] += 1
    return lg
lg = lg_of(used)
room = (1 << lg) // 2 + 1 - used                       # inserts the row still admits before it must double
k = keys_of(nb)
x = (k >> np.uint64(32)).astype(np.uint32)
pos = np.searchsorted(allk, k); pos[pos >= allk.size] = allk.size - 1
isnew = allk[pos] != k
newk = np.unique(k[isnew])
nrow = (newk >> np.uint64(32)).astype(np.uint32)
ri = np.searchsorted(ur, nrow); ri[ri >= ur.size] = ur.size - 1
known = ur[ri] == nrow
print('ops', B, 'ops on new keys', int(isnew.sum()), 'distinct new keys', newk.size, 'of them in new rows', int((~known).sum()))
# per existing row: new keys, ops
new_per_row = np.bincount(ri[known], minlength=ur.size)
xi = np.searchsorted(ur, x); xi[xi >= ur.size] = ur.size - 1
xk = ur[xi] == x
ops_per_row = np.bincount(xi[xk], minlength=ur.size)
pr = (prev_new >> np.uint64(32)).astype(np.uint32)
pi = np.searchsorted(ur, pr); pi[pi >= ur.size] = ur.size - 1
prev_per_row = np.bincount(pi[ur[pi] == pr], minlength=ur.size)
tot_ins = new_per_row.sum()
def share(mask, what):
    print('  %-58s rows %8d  inserts %9d = %5.1f %%' % (what, int(mask.sum()), int(new_per_row[mask].sum()), 100.0 * new_per_row[mask].sum() / tot_ins))
print('(b) inserts into existing rows: %d (rows with inserts: %d)' % (tot_ins, int((new_per_row > 0).sum())))
share(new_per_row <= room, 'room >= true new keys (oracle)')
share(ops_per_row <= room, 'room >= ops naming the row in the batch (histogram bound)')
share(2 * ops_per_row <= room, 'room >= 2 x ops naming the row')
for f in (2, 4, 8):
    share(f * prev_per_row + 8 <= room, 'room >= %d x inserts of the previous batch + 8 (estimate)' % f)
for s in (8, 10, 12, 15):
    m = lg >= s
    share(m, 'rows of >= 2^%d cells' % s)
tile_all = (np.arange(B, dtype=np.uint64) // np.uint64(2048))
print('    a per-batch row histogram (the bound of the second line) costs one atomic per distinct (tile, row) pair: %d' %
      np.unique(tile_all << np.uint64(32) | x.astype(np.uint64)).size)
grow = new_per_row > room
print('(a) rows that must double in this batch: %d; their inserts %d; ops deferred at least once ~ inserts beyond room: %d' %
      (int(grow.sum()), int(new_per_row[grow].sum()), int((new_per_row[grow] - room[grow]).sum())))
for lo, hi in ((4, 9), (10, 13), (14, 31)):
    m = grow & (lg >= lo) & (lg <= hi)
    print('    old size 2^%d..2^%d: rows %6d  cells %9d  inserts beyond room %8d' % (lo, hi, int(m.sum()), int((1 << lg[m]).sum()), int((new_per_row[m] - room[m]).sum())))
twice = new_per_row > room + (1 << lg) // 2
print('    rows that double twice or more: %d' % int(twice.sum()))
# (c) entries of the LDS fold by their op count inside the tile
tile = (np.arange(B, dtype=np.uint64) // np.uint64(2048))
tk = tile * np.uint64(1 << 40) ^ (k * np.uint64(0x9E3779B97F4A7C15))      # (a hash of (tile, key): collisions are negligible)
u, c = np.unique(tk, return_counts=True)
print('(c) (tile, key) entries: %d; with ONE op in their tile: %d = %.1f %% of the entries, %.1f %% of the ops' %
      (u.size, int((c == 1).sum()), 100.0 * (c == 1).sum() / u.size, 100.0 * (c == 1).sum() / B))
for lim in (2, 4, 8):
    print('    entries with >= %d ops: %d (%.1f %% of the entries, %.1f %% of the ops)' % (lim, int((c >= lim).sum()), 100.0 * (c >= lim).sum() / u.size, 100.0 * c[c >= lim].sum() / B))
# keys by their number of TILES in the batch: the entries a cross-tile fold would remove
uk, ck = np.unique(k, return_counts=True)
ek = np.unique(np.stack([tile, k], 1), axis=0)[:, 1]
_, tiles_per_key = np.unique(ek, return_counts=True)
print('    distinct keys %d; keys in ONE tile: %d; entries of keys in >= 2 tiles: %d' % (uk.size, int((tiles_per_key == 1).sum()), int(tiles_per_key[tiles_per_key >= 2].sum())))
