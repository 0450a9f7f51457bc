import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from graphsage_amd import GraphSAGE, TrainStep
from graphsage_amd.synth import cooccurrence_batch, memory_features, powerlaw_graph
n=10_000_000
ei=powerlaw_graph(n,20,seed=42,device="cuda"); x=memory_features(n,seed=42,device="cuda")
torch.manual_seed(0)
m=GraphSAGE(7,64,64,2,0.3).cuda(); ts=TrainStep(m,x,ei)
bs=[cooccurrence_batch(ei,n,1<<20,5,seed=s) for s in range(4)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(steps):
    st=ts.step(*bs[i%4])
    if i in (5, 100, steps - 1) or i % 1000 == 999:
        torch.cuda.synchronize()
        print(i, "alloc GB", round(torch.cuda.memory_allocated()/2**30,2), "peak GB", round(torch.cuda.max_memory_allocated()/2**30,2), "reserved GB", round(torch.cuda.memory_reserved()/2**30,2), "loss", float(st[0]), flush=True)
