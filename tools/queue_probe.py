"""developer tool: which streams share a hardware queue?  A long spin kernel on stream A, then a short one on stream B: in a
shared (in-order) hardware queue B ends after A, in different queues long before.  Prints the sharing pattern of the torch
stream pool against the default stream, and what a two-branch captured graph does (serialised = sum of its branches)."""
import sys, time
import torch
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
main = torch.cuda.current_stream(dev)
SPIN = int(2.0e6)        # ~1 ms at 2 GHz


def shares(a, b):
    torch.cuda.synchronize()
    ea1, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(SPIN)
        ea1.record(a)
    with torch.cuda.stream(b):
        torch.cuda._sleep(1000)
        eb1.record(b)
    torch.cuda.synchronize()
    return e0.elapsed_time(eb1) >= 0.8 * e0.elapsed_time(ea1)


def graph_time(n=5):
    """two-branch graph: spin on the launch stream, spin on a forked stream; ms per replay (serialised: 2x)"""
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        torch.cuda._sleep(SPIN)
        with torch.cuda.stream(side):
            torch.cuda._sleep(SPIN)
        cur.wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return g, (time.perf_counter() - t0) / n * 1e3


x = torch.zeros(1, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(SPIN); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3
print('one spin: %.2f ms' % one)
streams = [torch.cuda.Stream(dev) for _ in range(12)]
for s_ in streams:                      # first submission (creates / binds the hardware queue) outside the probes
    with torch.cuda.stream(s_):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(SPIN); torch.cuda.synchronize(); one = (time.perf_counter() - t0) * 1e3
print('one spin (warm): %.2f ms' % one)
print('pool streams sharing the default stream\'s queue:', [int(shares(main, s)) for s in streams])
print('pool stream i sharing pool stream 0\'s queue:   ', [int(shares(streams[0], s)) for s in streams])
keep = []
for i in range(8):
    g, ms = graph_time()
    keep.append(g)
    print('graph %d: %.2f ms per replay (%s)' % (i, ms, 'SERIALISED' if ms > 1.6 * one else 'parallel'))
