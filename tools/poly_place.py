import numpy as np, collections, sys
h=np.fromfile(sys.argv[1],dtype=np.uint64)
nw=int(sys.argv[2])
w=h[4096:4096+4*nw].reshape(nw,4)
def place(v):
    hw, xcc = int(v)&0xFFFFFFFF, (int(v)>>32)&7
    return (xcc,(hw>>13)&7,(hw>>12)&1,(hw>>8)&15,(hw>>4)&3)
wp=[place(v) for v in w[:,2]]
t0=int(w[:,0][w[:,0]>0].min())
ws=(w[:,0].astype(np.int64)-t0)*0.01; we=(w[:,1].astype(np.int64)-t0)*0.01
c=collections.Counter(wp)
print("simds used", len(c), "waves per simd:", sorted(collections.Counter(c.values()).items()))
cu=collections.Counter(p[:4] for p in wp)
print("CUs", len(cu), "waves per CU:", sorted(collections.Counter(cu.values()).items()))
for n in sorted(set(c.values())):
    d=[we[i]-ws[i] for i in range(nw) if c[wp[i]]==n]
    print(f" simds with {n} waves: durations med {np.median(d):.1f} max {np.max(d):.1f}")
print("pairs on same simd:", sum(1 for i in range(0,nw,2) if wp[i]==wp[i+1]), "of", nw//2)
print([wp[i] for i in range(16)])
