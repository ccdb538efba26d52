"""developer tool: how does the HIP runtime place NEW streams on hardware queues?  (mh_stream_create / mh_streams_share_queue)
1. classes of 10 new streams + the default stream; 2. create X, note its class, destroy it, create Y: same class (least-used
queue) or the next one (round robin)?  3. a two-branch graph instantiated right after such a probe: does its side branch share
the queue of a stream created right before it / of the default stream?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'scene-aware-3d-multi-human_amd')]
import torch
from mhhip import _lib
L = _lib.lib()
torch.zeros(1, device='cuda:0'); torch.cuda.synchronize()
pool = [torch.cuda.Stream() for _ in range(3)]           # (the torch pool of 32 + 32 streams exists from here on)
for s in pool:
    with torch.cuda.stream(s): torch.cuda._sleep(100)
torch.cuda.synchronize()


def new():
    p = ctypes.c_void_p()
    _lib.check(L.mh_stream_create(ctypes.byref(p)))
    return p.value


def share(a, b):
    r = ctypes.c_int(0)
    _lib.check(L.mh_streams_share_queue(a, b, 200.0, ctypes.byref(r)))
    return r.value


def classes(streams):
    cls = []
    for i, s in enumerate(streams):
        for j in range(i):
            if share(streams[j], s):
                cls.append(cls[j]); break
        else:
            cls.append(max(cls) + 1 if cls else 0)
    return cls

main = torch.cuda.current_stream().cuda_stream          # 0 = the default stream
xs = [new() for _ in range(10)]
t0 = time.perf_counter(); c = classes([main] + xs); dt = time.perf_counter() - t0
print('classes [default, 10 new streams]:', c, ' (%.1f ms for %d probes)' % (dt * 1e3, 55))
print('pool streams vs default:', [share(main, s.cuda_stream) for s in pool], ' vs new[0]:', [share(xs[0], s.cuda_stream) for s in pool])
ref = [main] + xs
def cls_of(s):
    for r, k in zip(ref, c):
        if share(r, s): return k
    return -1
for rep in range(6):
    x = new(); kx = cls_of(x); _lib.check(L.mh_stream_destroy(x))
    y = new(); ky = cls_of(y)
    print('create X -> class %d, destroy, create Y -> class %d' % (kx, ky))
    _lib.check(L.mh_stream_destroy(y))


def graph_side_class(spin=int(2e6)):
    """instantiate a two-branch graph; which class does its internal side stream share?  (replay beside a long spin on one
    representative stream per class: the replay is stretched when the side branch -- or the chain -- waits behind it)"""
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        torch.cuda._sleep(100000)
        with torch.cuda.stream(side):
            torch.cuda._sleep(100000)
        cur.wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); base = time.perf_counter() - t0
    out = []
    reps = {}
    for r, k in zip(ref, c):
        reps.setdefault(k, r)
    for k, r in sorted(reps.items()):
        if r == main:
            continue
        ext = torch.cuda.ExternalStream(r)
        torch.cuda.synchronize()
        with torch.cuda.stream(ext):
            torch.cuda._sleep(spin)
        t0 = time.perf_counter(); g.replay(); torch.cuda.current_stream().synchronize(); dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        out.append((k, round(dt / base, 1)))
    return g, base * 1e3, out

keep = []
for rep in range(6):
    x = new(); kx = cls_of(x); _lib.check(L.mh_stream_destroy(x))
    g, base, out = graph_side_class()
    keep.append(g)
    print('new stream X -> class %d (destroyed); graph instantiated next: %.3f ms alone (2 branches of ~0.05 ms: serialised = 2x); '
          'replay time / alone beside a spin on class k: %s' % (kx, base, out))
