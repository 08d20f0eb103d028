import sys
sys.path[:0]=["/root/repo","/root/repo/vectordb-retrieval_amd"]
import numpy as np, vdbhip
from bench import make_data
X,Q,k,m=make_data("sift1m",0)
idx=vdbhip.IVFFlatIndex(128,1024,m,0); idx.train(X, niter=25, seed=1234, max_points_per_centroid=256); idx.add(X)
c=np.bincount(idx.assignment(),minlength=1024)
print("list sizes: min",c.min(),"median",int(np.median(c)),"p90",int(np.percentile(c,90)),"p99",int(np.percentile(c,99)),"max",c.max())
print("top10",np.sort(c)[-10:])
