"""Statistics of the dropout hash of bsi_amd/csrc/common.h (drop_quad): drop rates, dispersion, field correlations, bit balance,
avalanche per input bit, chi-square of the 16-bit fields over sequential inputs.  numpy restatement of the device function."""
import numpy as np
M32=0xffffffff
def mad24(a,b,c):
    return (( (a & 0xffffff).astype(np.uint64) * np.uint64(b & 0xffffff)) + c.astype(np.uint64)) & np.uint64(M32)
def mix_old(x):
    x = x.astype(np.uint64)
    x ^= x>>np.uint64(16); x = (x*np.uint64(0x7feb352d))&np.uint64(M32); x ^= x>>np.uint64(15); x=(x*np.uint64(0x846ca68b))&np.uint64(M32); x^= x>>np.uint64(16)
    return x
def quad(x):  # drop_quad of bsi_amd/csrc/common.h
    x = x.astype(np.uint64)
    t = mad24(x, 0x9E3779, x>>np.uint64(12))
    t = t ^ (t>>np.uint64(15))
    t = mad24(t, 0x85EBCB, t>>np.uint64(10))
    h1 = t ^ (t>>np.uint64(14))
    u = mad24(h1, 0xC2B2AF, h1>>np.uint64(6))
    h2 = u ^ (u>>np.uint64(13))
    return h1&np.uint64(M32), h2&np.uint64(M32)
rng=np.random.default_rng(0)
# rows: rowh = mix_old(row + s0) ^ s1 ; cols 0..255 ; 4096 rows
s0, s1 = 0x1234567, 0x9abcdef0
rows = np.arange(8192, dtype=np.uint64)
rowh = mix_old((rows + s0) & M32) ^ np.uint64(s1)
quads = np.arange(64, dtype=np.uint64)
x = (quads[None,:] ^ rowh[:,None]) & np.uint64(M32)
h1,h2 = quad(x)
f = np.stack([h1&0xffff, h1>>16, h2&0xffff, h2>>16], axis=-1).reshape(8192, 256).astype(np.float64)
for p in (0.05, 0.1, 0.5):
    thr = int(p*65536+0.5)
    keep = (f >= thr)
    print("p",p,"drop rate", 1-keep.mean(), "expected", thr/65536, "std of row drop counts", (~keep).sum(1).std(), "binomial", np.sqrt(256*p*(1-p)), "col", (~keep).sum(0).std(), np.sqrt(8192*p*(1-p)))
# correlations between fields / neighbours
u = f/65536
def corr(a,b): return np.corrcoef(a.ravel(), b.ravel())[0,1]
print("adjacent col corr", corr(u[:,:-1],u[:,1:]), "col+2", corr(u[:,:-2],u[:,2:]), "col+4", corr(u[:,:-4],u[:,4:]), "row+1", corr(u[:-1],u[1:]))
# bit balance
for name,h in (("h1",h1),("h2",h2)):
    bits = ((h[...,None]>>np.arange(32,dtype=np.uint64))&np.uint64(1)).mean((0,1))
    print(name,"bit means min/max", bits.min(), bits.max())
# avalanche: flip each input bit, count output flips
x0 = rng.integers(0, 2**32, size=20000, dtype=np.uint64)
a1,a2 = quad(x0)
worst=[]
for b in range(32):
    b1,b2 = quad(x0 ^ np.uint64(1<<b))
    d1 = np.array([bin(int(v)).count("1") for v in (a1^b1)[:2000]]).mean()
    d2 = np.array([bin(int(v)).count("1") for v in (a2^b2)[:2000]]).mean()
    worst.append((b, round(d1,1), round(d2,1)))
print(worst)
# chi-square of 16-bit fields over sequential quads for a fixed rowh (structured input)
xs = (np.arange(1<<20, dtype=np.uint64) ^ np.uint64(0xdeadbeef))
g1,g2 = quad(xs)
for nm,fld in (("h1lo",g1&0xffff),("h1hi",g1>>16),("h2lo",g2&0xffff),("h2hi",g2>>16)):
    cnt = np.bincount((fld>>8).astype(np.int64), minlength=256)
    chi = ((cnt-cnt.mean())**2/cnt.mean()).sum()
    print(nm, "chi2(255 dof)", round(chi,1))
