"""CPU census of what the LDS tile fold of k_apply_agg can and cannot remove (no GPU needed; ~3 min, ~8 GB):
the config-2 stream up to batch `nb`, then for batch nb: distinct (tile, key) entries vs distinct keys, new keys, and how
many `used` tickets a per-(tile, row) aggregation would save.  python tools/probe/tile_fold_census.py [nb]
Round 3, nb = 15:  16.8 M ops -> 15.0 M (tile, key) entries (7.0 M distinct keys in the batch); 3.76 M ops on new keys =
3.76 M entries; a ticket per (tile, row) instead of per new key: 3.56 M -- 5 % fewer (2.79 M of 2.80 M in small rows,
0.77 M of 0.96 M in rows of >= 2^15 cells)."""
import numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from libsmatrix_amd import Stream
B=1<<24
nb=int(sys.argv[1]) if len(sys.argv)>1 else 15
gen=Stream("zipf",12345,1000000,1.1,1)
t=time.time()
keys=[]
for b in range(nb):
    x,y=gen.fill(b*B,B)
    k=(x.astype(np.uint64)<<np.uint64(32))|y.astype(np.uint64)
    keys.append(np.unique(k))
    print('batch',b,time.time()-t,flush=True)
allk=np.unique(np.concatenate(keys)); del keys
print('nnz before',allk.size)
rows_before=(allk>>np.uint64(32)).astype(np.uint32)
ur,cnt=np.unique(rows_before,return_counts=True)
# row size lg from n distinct keys: smallest 16*2^k with n <= 8*2^k+1
def lg_of(n):
    lg=np.full(n.shape,4,np.int64)
    while True:
        m=n>(1<<lg)//2+1
        if not m.any(): break
        lg[m]+=1
    return lg
lgs=lg_of(cnt)
x,y=gen.fill(nb*B,B)
k=(x.astype(np.uint64)<<np.uint64(32))|y.astype(np.uint64)
pos=np.searchsorted(allk,k); pos[pos>=allk.size]=allk.size-1
isnew=allk[pos]!=k
print('ops on new keys',isnew.sum())
tile=np.arange(B)//2048
# distinct (tile,key) entries
tk=np.unique(np.stack([tile[isnew].astype(np.uint64),k[isnew]],1),axis=0)
print('distinct (tile,newkey) entries',tk.shape[0], 'distinct new keys', np.unique(k[isnew]).size)
rows=(tk[:,1]>>np.uint64(32)).astype(np.uint32)
ri=np.searchsorted(ur,rows); ri[ri>=ur.size]=ur.size-1
known=ur[ri]==rows
rlg=np.where(known,lgs[ri],4)
big=rlg>=15
print('entries in big rows',big.sum(),'small',(~big).sum(), 'new rows', (~known).sum())
tr=np.unique(np.stack([tk[:,0],rows.astype(np.uint64)],1),axis=0)
print('distinct (tile,row) pairs overall',tr.shape[0])
trb=np.unique(np.stack([tk[big,0],rows[big].astype(np.uint64)],1),axis=0)
trs=np.unique(np.stack([tk[~big,0],rows[~big].astype(np.uint64)],1),axis=0)
print('big: entries',big.sum(),'-> (tile,row) pairs',trb.shape[0]); print('small: entries',(~big).sum(),'-> pairs',trs.shape[0])
# all distinct (tile,key) entries
tka=np.unique(tile.astype(np.uint64)*np.uint64(1<<40) ^ (k*np.uint64(0x9E3779B97F4A7C15)))
print('distinct (tile,key) all ~',tka.size,'distinct keys in batch',np.unique(k).size)
