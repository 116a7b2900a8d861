"""Does a hipGraph replay of the headline step (counter reset + K1 + fallback + robust pass) beat four eager launches?
python scripts/dev/graph_vs_eager.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np, torch
from feos_torch_amd import native
from feos_torch_amd.synthetic import pure_batch
n = 10_000_000
P, T = pure_batch(n)
dev = torch.device("cuda:0")
par, tem = torch.from_numpy(P).to(dev), torch.from_numpy(T).to(dev)
plan = native.PureVlePlan(n, dev)
for _ in range(5): plan.run(par, tem)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    plan.run(par, tem)
torch.cuda.current_stream(dev).wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    plan.run(par, tem)
def timed(fn, steps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
for rnd in range(3):
    print(f"eager {timed(lambda: plan.run(par, tem)):.4f} ms/step   graph replay {timed(graph.replay):.4f} ms/step")
