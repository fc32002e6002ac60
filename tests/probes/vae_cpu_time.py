import sys,time,torch; sys.path.insert(0,'.')
from oracle import wan_vae_oracle as V
cfg=V.VAEConfig(dim=96); sd=V.synth_state_dict(cfg,"t")
for thr in (32, 64, 128):
    torch.set_num_threads(thr)
    z=torch.randn(16,2,15,26)
    t0=time.time(); V.vae_decode(sd,cfg,z[:, :1]); t1=time.time()-t0
    t0=time.time(); V.vae_decode(sd,cfg,z); t2=time.time()-t0
    print(thr, "first", round(t1,1), "two", round(t2,1), flush=True)
