import numpy as np, sys
sys.path.insert(0,'/root/repo')
import grakel_amd as gk
from grakel_amd.synthetic import nci1_like
from grakel_amd.batch import sp_batch_from_input
from grakel_amd.engine import get_engine
from oracle import grakel_oracle as O
G = nci1_like(300,0,True)
Ko = O.SPOracle().fit_transform(G)
eng = get_engine(); gb,_ = sp_batch_from_input(G, True)
eq = np.array_equal
ref = None
for it in range(6):
    db = eng.upload(gb); pb = eng.sp_build(db, None, True)
    feat = eng.features(pb, 1); K1 = eng.gram(feat); sk = eng.selfk(feat); P = eng.debug_phi(feat)
    Kh = P @ P.T; np.fill_diagonal(Kh, sk)
    print(it, "K1", eq(K1,Ko), "host-gram-of-phi", eq(Kh,Ko), "gemm==host", eq(K1,Kh), "phi nnz", int((P!=0).sum()), "phi sum", P.sum(), "colsum>0", int((P.sum(0)>0).sum()), P.shape)
    colsig = np.sort((P*np.arange(1,301)[:,None]).sum(0))
    if ref is None: ref = colsig
    else: print("    same column multiset as run0:", eq(colsig, ref))
    del feat, pb, db
