import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from redmax_amd import BatchSim
from redmax_amd.scenes import sceneTree, sceneChain
from oracle import oracle as orc
def states(sc,B):
    qs,_=sc.getQ(); q=np.empty((B,sc.nr)); qd=np.empty((B,sc.nr))
    for i in range(B):
        rng=np.random.default_rng(20240+i); q[i]=qs+rng.uniform(-0.05,0.05,sc.nr); qd[i]=rng.uniform(-0.1,0.1,sc.nr)
    return q,qd
for sc in (sceneTree(64),):
    sc.init(); B=3; q,qd=states(sc,B)
    sims={}
    for mode in ("0","1"):
        os.environ["RMX_W2"]=mode
        s=BatchSim(sc,batch=B); s.set_state(q,qd); sims[mode]=s
    os_=[orc.Oracle(sc.desc()) for b in range(B)]
    for b in range(B): os_[b].set_state(q[b],qd[b])
    # compare H,g eval first
    for k in range(6):
        r={}
        for mode in ("0","1"):
            os.environ["RMX_W2"]=mode
            out=sims[mode].step_bdf1(1,h=1e-2,stats=True)
            r[mode]=(sims[mode].get_state()[0],out)
        oi=[]
        for b in range(B):
            st=os_[b].step_bdf1(1e-2,1); oi.append(st.newton_iters)
        qo=np.array([os_[b].get_state()[0] for b in range(B)])
        print("step",k,"iters w1",r["0"][1]["newton_iters"],"w2",r["1"][1]["newton_iters"],"oracle",oi,"status",r["0"][1]["status"],r["1"][1]["status"],
              "rel w2-w1 %.2e"%(np.linalg.norm(r["1"][0]-r["0"][0])/np.linalg.norm(r["0"][0])),"w1-orc %.2e"%(np.linalg.norm(r["0"][0]-qo)/np.linalg.norm(qo)),"w2-orc %.2e"%(np.linalg.norm(r["1"][0]-qo)/np.linalg.norm(qo)), "halv", r["0"][1]["ls_halvings"], r["1"][1]["ls_halvings"])
