PGF_LOOKAHEAD=1 PGF_COHERENT=3 TAG=la python tools/dbg_la2.py
TAG=serial python tools/dbg_la2.py
python - <<PY
import numpy as np, collections
a=np.load("gpurun_out/factor_la.npy"); b=np.load("gpurun_out/factor_serial.npy")
d=np.abs(a-b); bad=np.argwhere(d>1e-9*np.abs(b).max())
print("num bad", len(bad))
if len(bad):
    rows=bad[:,0]; cols=bad[:,1]
    print("row range", rows.min(), rows.max(), "col range", cols.min(), cols.max())
    cb=collections.Counter((c//64) for c in cols)
    print("bad per col block (first 6):", sorted(cb.items())[:6])
    c0=min(cb)*64
    sel=bad[(cols>=c0)&(cols<c0+64)]
    rb=collections.Counter((r//64) for r in sel[:,0])
    print("first bad col block", c0, "bad row blocks:", sorted(rb.items())[:24])
    print("first bad cols in that block:", sorted(set(sel[:,1]))[:10])
    r0=sel[:,0].min(); print("sample diffs", [(int(r),int(c),float(a[r,c]),float(b[r,c])) for r,c in sel[:3]])
PY
rm -f gpurun_out/factor_*.npy
