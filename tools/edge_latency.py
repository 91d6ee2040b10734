#!/usr/bin/env python3
"""What does a dependency edge cost on this runtime?  Chains of N tiny dependent kernels (one float each), timed end to end:
  same     : all on one stream                                   (in-order queue: barrier bit between packets)
  pingpong : alternating between two streams, an event per hop    (cross-queue signal per edge)
  fork     : main chain on one stream, every link also forks a side kernel on a second stream joined one link later
each issued eagerly (host far ahead: measured after a warm pass) and as a replayed hipGraph.    python tools/edge_latency.py"""
import sys

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None   # e.g. "pingpong:graph,fork:graph" (one case per process: a
#                                                                 failing capture takes the process down)
dev = "cuda:0"
x = torch.zeros(64, device=dev)
y = torch.zeros(64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def same():
    for _ in range(N):
        x.add_(1.0)


def pingpong():
    cur, other = s1, s2
    for _ in range(N):
        with torch.cuda.stream(cur):
            x.add_(1.0)
        other.wait_stream(cur)
        cur, other = other, cur
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)


def fork():
    main = torch.cuda.current_stream()
    for _ in range(N):
        x.add_(1.0)
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            y.add_(1.0)
        main.wait_stream(s2)     # (joined before the NEXT link: the side kernel overlaps nothing here, worst case)


def timed(fn, graph):
    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        s1.wait_stream(cs)
        s2.wait_stream(cs)
        if graph:
            g = torch.cuda.CUDAGraph()
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cs):
                if fn is pingpong:
                    s1.wait_stream(cs)
                fn()
            run = g.replay
        else:
            run = fn
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * N)


for name, fn in (("same", same), ("pingpong", pingpong), ("fork", fork)):
    for graph in (False, True):
        if ONLY is not None and f"{name}:{'graph' if graph else 'eager'}" not in ONLY:
            continue
        try:
            us = timed(fn, graph)
            print(f"{name:9s} {'graph' if graph else 'eager':6s} {us:7.2f} us per link ({N} links)")
        except Exception as e:  # noqa: BLE001
            print(f"{name:9s} {'graph' if graph else 'eager':6s} failed: {type(e).__name__}: {str(e)[:120]}")


# ---- what does the boundary between two replays cost?  The same chain as ONE graph of N links, as two graphs of N / 2 links
#      replayed back to back, and with a tiny eager kernel between the two (the step's out-of-graph schedule fills / copies)
if ONLY is None or "boundary" in ONLY:
    cs = torch.cuda.Stream()
    z = torch.zeros(64, device=dev)
    with torch.cuda.stream(cs):
        def chain(k):
            for _ in range(k):
                x.add_(1.0)
        graphs = {}
        for k in (N, N // 2):
            g = torch.cuda.CUDAGraph()
            chain(k)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cs):
                chain(k)
            graphs[k] = g

        def t(run, reps=20):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps

        one = t(lambda: graphs[N].replay())
        two = t(lambda: (graphs[N // 2].replay(), graphs[N // 2].replay()))
        mid = t(lambda: (graphs[N // 2].replay(), z.add_(1.0), graphs[N // 2].replay(), z.add_(1.0)))
        print(f"boundary: one graph of {N} links {one:8.1f} us; two of {N // 2}: {two:8.1f} us (+{two - one:.1f} for one more boundary); "
              f"with an eager kernel after each: {mid:8.1f} us (+{(mid - two) / 2:.1f} per eager kernel between replays)")
