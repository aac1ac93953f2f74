import sys, numpy as np, torch
sys.path.insert(0, '.')
import instancefusion_amd as ifx
from instancefusion_amd import synth
W,H=640,480; K=dict(fx=528.0,fy=528.0,cx=320.0,cy=240.0)
st=synth.make_stream(6,W,H,noise=True,loop_len=90,**K)
m=synth.make_map(1_000_000, st["scene"], st["poses_world"][0], 1000)
hs=[]
for hot in (0,1):
    g=ifx.ElasticFusion(w=W,h=H,max_surfels=2_000_000,**K)
    g.set_option("hot_records",hot)
    g.processFrame(st["rgb"][0],st["depth"][0])
    g.upload(m); g.set_pose(st["poses"][0],1000); g.combined_predict(st["poses"][0],1000,1000)
    hs.append(g)
for i in range(1,5):
    ps=[g.processFrame(st["rgb"][i],st["depth"][i]) for g in hs]
    print(i,'pose equal',np.array_equal(ps[0],ps[1]), 'count',hs[0].count,hs[1].count)
    for name in ("pred_vertex","pred_normal","pred_image","pred_time","fill_vertex","ids_after"):
        a,b=hs[0].image(name),hs[1].image(name)
        d=(a!=b)
        if d.any():
            idx=np.argwhere(d.reshape(H,W,-1).any(2))
            print('   ',name,'differs at',len(idx),'pixels, first',idx[:3].tolist())
m0,m1=hs[0].download(),hs[1].download()
for k in m0: print(k, np.array_equal(m0[k],m1[k]), (m0[k]!=m1[k]).any(axis=1).sum() if m0[k].shape==m1[k].shape else (m0[k].shape,m1[k].shape))
