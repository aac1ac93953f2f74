import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import instancefusion_amd as ifx
import oracle_lib as orc
from instancefusion_amd import synth
orc.build(); orc.set_threads(orc.usable_cores())
Wb,Hb,NFb=640,480,24; Kb=dict(fx=528.0,fy=528.0,cx=320.0,cy=240.0); seed,motion=302,"fast"
scene=synth.Scene(seed)
st=synth.make_stream_from_poses(synth.trajectory_profile(motion,NFb,seed),scene,Wb,Hb,noise_seed=seed+1,**Kb)
d_rgb=torch.from_numpy(st["rgb"]).cuda(); d_dep=torch.from_numpy(st["depth"].view(np.int16)).cuda(); torch.cuda.synchronize()
for resident in (1,0):
    g=ifx.ElasticFusion(w=Wb,h=Hb,max_surfels=2_000_000,confidence=3.0,**Kb); o=orc.Oracle(w=Wb,h=Hb,max_surfels=2_000_000,confidence=3.0,**Kb); inst=ifx.InstanceFusion(g)
    for i in range(NFb):
        if resident:
            if i+1<NFb: g.hint_next_frame_device(d_rgb[i+1].data_ptr(),d_dep[i+1].data_ptr())
            g.enqueue_frame_device(d_rgb[i].data_ptr(),d_dep[i].data_ptr(),i); inst.whetherDoSegmentation(100+i)
        else: g.processFrame(st["rgb"][i],st["depth"][i])
        o.process_frame(st["rgb"][i],st["depth"][i])
    i=NFb-1
    masks,cls=synth.canned_masks(st["obj"][i],scene)
    vg0,vo0=g.download()["votes"].copy(),o.download()["votes"].copy()
    print('resident',resident,'votes equal before call',np.array_equal(vg0,vo0), 'count', g.count, o.count)
    # NOTE download compacts; redo the last frame state is unchanged otherwise
    if resident: inst.ProcessSegmentation(None,None,masks,cls,i,superpixels=False)
    else: inst.ProcessSegmentation(st["rgb"][i],st["depth"][i],masks,cls,i,superpixels=False)
    o.process_segmentation(st["rgb"][i],st["depth"][i],masks,cls,i,flags=0)
    vg,vo=g.download()["votes"],o.download()["votes"]
    d=(vg!=vo)
    print('   after call equal',np.array_equal(vg,vo),'rows differing',d.any(1).sum(),'table',inst.getInstanceTable()[:12], o.instance_table()[:12])
    lg,lo=inst.labels(),o.labels()
    print('   labels shapes',lg.shape,lo.shape,'equal',np.array_equal(lg,lo), 'diff count', (lg!=lo).sum() if lg.shape==lo.shape else None, 'gpu count', g.count, 'slots', g.slots)
    if lg.shape==lo.shape and (lg!=lo).any():
        w=np.argwhere(lg!=lo)[:,0]; print('   label diffs at',w[:8],'gpu',lg[w[:8]],'orc',lo[w[:8]])
    if d.any():
        r,c=np.argwhere(d)[0]; print('   first diff row',r,'col',c,'gpu',vg[r,c],'orc',vo[r,c],'before',vg0[r,c] if r<len(vg0) else None, vo0[r,c])
        rows=np.argwhere(d.any(1))[:,0]; print('   rows range',rows.min(),rows.max(),'cols',np.unique(np.argwhere(d)[:,1])[:10])
    g.close(); o.close()
