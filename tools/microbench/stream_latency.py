#!/usr/bin/env python3
"""What does a cross-stream dependency cost on this stack?  N short kernels (~20 us each, long enough for the host to
stay ahead) issued (A) on one stream, (B) on one stream with an event recorded after each, (C) alternating between two
streams with an event wait per hop (ping-pong), (D) the backward's pattern: main kernel, fork a companion kernel that
waits for it, next main kernel (no wait), join at the end.  Reported: GPU time per kernel beyond the kernel itself."""
import time

import torch

dev = torch.device("cuda:0")
x = torch.zeros(8 << 20, device=dev)  # 32 MB: add_ ~ 15-20 us
y = torch.zeros(8 << 20, device=dev)
N = 400
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best / N * 1e6


def a():
    with torch.cuda.stream(s1):
        for _ in range(N):
            x.add_(1.0)


def b():
    with torch.cuda.stream(s1):
        for _ in range(N):
            x.add_(1.0)
            e = torch.cuda.Event()
            e.record(s1)


def c():
    cur, oth = s1, s2
    for _ in range(N):
        with torch.cuda.stream(cur):
            x.add_(1.0)
        e = torch.cuda.Event()
        e.record(cur)
        oth.wait_event(e)
        cur, oth = oth, cur


def d():
    for _ in range(N // 2):
        with torch.cuda.stream(s1):
            x.add_(1.0)
        e = torch.cuda.Event()
        e.record(s1)
        s2.wait_event(e)
        with torch.cuda.stream(s2):
            y.add_(1.0)
    e = torch.cuda.Event()
    e.record(s2)
    s1.wait_event(e)


for f in (a, b, c, d):
    f()
base = run(a)
print("A one stream:                         %.2f us per kernel" % base)
print("B one stream + event record each:     %.2f us per kernel (+%.2f)" % (run(b), run(b) - base))
print("C two streams ping-pong (wait each):  %.2f us per kernel (+%.2f per hop)" % (run(c), run(c) - base))
print("D main + forked companion kernels:    %.2f us per kernel pair-half (two streams may overlap: < A means overlap)" % run(d))
# the same with tiny kernels: is the host the limit?
x2 = torch.zeros(1024, device=dev)


def tiny():
    with torch.cuda.stream(s1):
        for _ in range(N):
            x2.add_(1.0)


tiny()
print("E tiny kernels, one stream (host launch rate): %.2f us per launch" % run(tiny))
