#!/usr/bin/env python3
"""The same destroy -> recapture sequence through torch.cuda.CUDAGraph (what insv2v/inference.py uses): a captured chain of torch ops with
a fork / join over three side streams, replayed, the graph object deleted (hipGraphExecDestroy + release of its private pool), the same
shapes captured again and replayed.  mode (argv[1]): keep = the dead graphs stay referenced (the product's park), destroy = deleted,
destroy_empty = deleted + torch.cuda.empty_cache() before the recapture (the variant that crashed most reliably in round 3)."""
import gc
import sys
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "destroy"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda:0"
w = [torch.randn(2048, 2048, device=dev, dtype=torch.float16) for _ in range(3)]
x = torch.randn(3, 4096, 2048, device=dev, dtype=torch.float16)
streams = [torch.cuda.Stream() for _ in range(3)]
park = []


def chain():
    main = torch.cuda.current_stream()
    outs = [None] * 3
    for b, st in enumerate(streams):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            h = x[b]
            for _ in range(20):
                h = torch.nn.functional.silu(h @ w[b]) * 0.01
            outs[b] = h
    for st in streams:
        main.wait_stream(st)
    return torch.cat(outs, 0)


for r in range(rounds):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = chain()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"round {r}: capture -> replay x3 ok, |out| = {out.float().abs().mean().item():.4f}", flush=True)
    if mode == "keep":
        park.append(g)
    del g, out
    gc.collect()
    if mode == "destroy_empty":
        torch.cuda.empty_cache()
print(f"mode {mode}: {rounds} rounds survived")
