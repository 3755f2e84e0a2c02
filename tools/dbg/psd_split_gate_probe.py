"""Round 6, table 2 of profiles/r06_psd_split_model.txt: the split flow of tools/dbg/psd_split_model.py with the refinement gate's off-norm condition
taken over the MIXED-sign entries only (the best case for sign-split sweeps): three consecutive calls per (start iteration, block).
    python tools/dbg/psd_split_gate_probe.py gpurun_out/psd_iterates2.npz"""
import sys, numpy as np
sys.path.insert(0,__file__.rsplit('/', 1)[0])
import psd_split_model as M
from psd_refine_proto import svec_to_sym, proj_exact, dk_map
z=np.load(sys.argv[1]); n=200; NP=208; NB=26
def project(A,V,real,relax):
    Ap=np.zeros((NP,NP)); Ap[:n,:n]=A
    S=V.T@Ap@V; S=0.5*(S+S.T); steps=0; how=""
    a=M.sort_by_sign(S,V,real)
    for trial in range(8):
        st=M.stats(S,real)
        if st["off"]<=M.TOL or (st["mix"]<=M.TOL and st["om"]<=M.OM_RELAXED): how+="C"; break
        offc = st["mix"] if relax else st["off"]
        if st["kf"]<=M.K_GATE and offc<=M.OFF_GATE and st["om"]<=M.OM_GATE and "R" not in how:
            Q=M.refine(S,real,st["K1"]); S=Q.T@S@Q; S=0.5*(S+S.T); V[:]=V@Q; how+="R"; continue
        if st["om"]<=0.25 and 0<a<NB: steps+=M.sweep(S,V,M.schedule_split(NB,a),False); how+="s"
        else: steps+=M.sweep(S,V,M.schedule_full(NB),True); how+="F"
    X=(V@dk_map(S.copy())@V.T)[:n,:n]
    st=M.stats(S,real)
    return X,steps,how,st
for k0 in (131,161,201,241):
  for bi in (0,2,4):
    A0=svec_to_sym(z["z_%d"%k0][bi]); w,U=np.linalg.eigh(A0); V=np.eye(NP); V[:n,:n]=U; real=np.arange(NP)<n
    out=[]
    for it in range(k0+1,k0+4):
        A=svec_to_sym(z["z_%d"%it][bi]); X,steps,how,st=project(A,V,real,True)
        err=np.linalg.norm(X-proj_exact(A))/np.linalg.norm(A)
        out.append("%s/%d err %.0e off %.1e om %.1e"%(how,steps,err,st['off'],st['om']))
    print(k0,bi," | ".join(out),flush=True)
