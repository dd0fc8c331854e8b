import sys, time, torch
sys.path.insert(0, '/root/repo')
import two_towers_overlords_amd as pkg
from two_towers_overlords_amd.training import TripletTrainer, TripletBatch
from two_towers_overlords_amd.engine import flat_param_slices
V,D,P,B,S=30522,384,384,2048,128
dev=torch.device('cuda')
table=(torch.randn(V,D)*0.02).to(dev)
params={n:(torch.rand(*sh)*0.1-0.05).to(dev) for n,_,sh in flat_param_slices(D,P)}
for overlap in (False, True):
  for mode in ('ids3','index'):
    tr=TripletTrainer(table=table, params={k:v.clone() for k,v in params.items()}, max_batch=B, max_seq_len=S, overlap_streams=overlap)
    ids=torch.randint(0,V,(3,B,S),dtype=torch.int32,device=dev)
    ni=((torch.arange(B)+1)%B).int().to(dev)
    b = TripletBatch(ids=ids) if mode=='ids3' else TripletBatch(ids=ids[:2].contiguous(), neg_index=ni)
    for rep in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter()
        n=200
        issued=0
        def feed():
            global issued
            while issued<n and len(tr._pending)<(2 if overlap else 1):
                tr.gather(b); issued+=1
        issued=0; feed()
        while tr._pending:
            tr.apply(None, issue_next=feed); feed()
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"overlap={overlap} mode={mode}: host {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
