#!/usr/bin/env python
"""What a hipGraph replay costs PER NODE on this stack ("B 16 eager 5.64 ms, captured 6.67 ms: the
explanation -- ~600 nodes pay a per-node cost -- is asserted, not measured").  N launches of a kernel that does next to nothing
(loans_axpby_f32 on 256 floats), laid out like a training step -- a main stream with two side streams that fork from it and
join it every `--period` launches -- (a) eager, three free-running streams, (b) captured once and replayed.  Reported: the
GPU-side time per launch (HIP events around the whole sequence, median of `--reps`), and the host's time to enqueue it.
Development tool; its output is committed as profiles/r5_b16_graph_nodes.txt.
usage: graph_nodes.py [--nodes 600] [--period 12] [--reps 30]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, default=600)
ap.add_argument('--period', type=int, default=12)
ap.add_argument('--reps', type=int, default=30)
args = ap.parse_args()

dev = torch.device('cuda', 0)
x = [torch.ones(256, device=dev) for _ in range(3)]
y = [torch.zeros(256, device=dev) for _ in range(3)]
main = torch.cuda.Stream(device=dev)
side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def sequence():
    """N tiny launches: every `period` of them one goes to each side stream (forked from main, joined back one period later)"""
    pending = []
    for i in range(args.nodes):
        k = i % args.period
        if k in (1, 2):                         # a "weight gradient" / "assessor chain" launch beside the main stream
            st = side[k - 1]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                ops.axpby(1.0, x[k], 0.5, y[k])
            pending.append(st)
        else:
            if k == 0:
                for st in pending:
                    main.wait_stream(st)
                pending = []
            with torch.cuda.stream(main):
                ops.axpby(1.0, x[0], 0.5, y[0])
    for st in pending:
        main.wait_stream(st)


def timed(fn):
    gpu, host = [], []
    for _ in range(args.reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            e0.record()
        h0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - h0) * 1e3)
        with torch.cuda.stream(main):
            e1.record()
        torch.cuda.synchronize()
        gpu.append(e0.elapsed_time(e1))
    return float(np.median(gpu)), float(np.median(host))


sequence()
torch.cuda.synchronize()
eager_gpu, eager_host = timed(sequence)

graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    sequence()
torch.cuda.synchronize()
with torch.cuda.graph(graph, stream=main):
    sequence()


def replay():
    with torch.cuda.stream(main):
        graph.replay()


replay()
graph_gpu, graph_host = timed(replay)

# one stream, no forks: the floor of a dependent launch chain, eager and captured
def chain():
    with torch.cuda.stream(main):
        for _ in range(args.nodes):
            ops.axpby(1.0, x[0], 0.5, y[0])


chain()
chain_gpu, chain_host = timed(chain)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=main):
    chain()


def replay2():
    with torch.cuda.stream(main):
        g2.replay()


replay2()
chain_graph_gpu, chain_graph_host = timed(replay2)

n = args.nodes
print('%d launches of a 256-float kernel, median of %d repetitions (GPU time by HIP events around the sequence, host = time to enqueue it)' % (n, args.reps))
print('three streams, fork / join every %d launches (the shape of a training step):' % args.period)
print('  eager     : GPU %.3f ms = %.2f us per launch; host %.3f ms = %.2f us per launch' % (eager_gpu, eager_gpu / n * 1e3, eager_host, eager_host / n * 1e3))
print('  hipGraph  : GPU %.3f ms = %.2f us per node;   host %.3f ms' % (graph_gpu, graph_gpu / n * 1e3, graph_host))
print('one stream, a dependent chain:')
print('  eager     : GPU %.3f ms = %.2f us per launch; host %.3f ms = %.2f us per launch' % (chain_gpu, chain_gpu / n * 1e3, chain_host, chain_host / n * 1e3))
print('  hipGraph  : GPU %.3f ms = %.2f us per node;   host %.3f ms' % (chain_graph_gpu, chain_graph_gpu / n * 1e3, chain_graph_host))
