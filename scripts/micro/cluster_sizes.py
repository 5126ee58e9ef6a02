import sys,os
sys.path.insert(0,'/root/repo')
import numpy as np, gulon_amd as g
n=10_000_000
dm=g.DeviceMatrix.synthetic(n,300,2,1234,1)
v=g.Vectors(dm,0,10)
km=g.KMeans.init(256,v,0)
a=km.par_assign(v)
c=np.bincount(a,minlength=256)
print("iter0 max",c.max(),"mean",c.mean(),"p90",np.percentile(c,90), "min", c.min())
km2=km.iterate(v,2)
a=km2.par_assign(v)
c=np.bincount(a,minlength=256)
print("iter2 max",c.max(),"mean",c.mean(),"p90",np.percentile(c,90), "min", c.min())
