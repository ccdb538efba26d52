"""Which hardware queue does the scene update run on?

A captured cycle (``SequenceEngine.cycle_graphed``) keeps up to three in-order queues busy at once: the graph's chain, the
graph's side branch -- on streams the HIP runtime creates when the graph is instantiated: they cannot be seen or chosen --
and the device-side scene update of reference optimizer.py:578-584 on a stream of ours.  The runtime multiplexes every stream
of the process onto a few hardware queues (4 unless GPU_MAX_HW_QUEUES says otherwise), and two streams on one hardware queue
run one after the other.  Measured on MI355X (tools/fit_cycles.py, round 6): the same ``fit(250)`` at C3 takes 0.81 ms per
cycle with the scene update on a queue of its own and 1.0 ms with it on the queue of the graph's side branch -- and with one
pooled torch stream for every engine of a process (rounds 2-5) which of the two a fit got depended on how many streams the
process had created before: the first optimiser of a process was fast, the second slow (new streams are dealt out over the
hardware queues in turn, tools/queue_policy.py; creating and destroying streams to steer that costs 0.2 + 0.5 ms apiece).

So the queue is CHOSEN, per graph, by looking:

* ``lanes``: once per process, one stream per hardware queue (streams are created until every new one shares a queue with an
  earlier one; ``mh_streams_share_queue``: a spin kernel on one stream, an empty kernel on the other -- in-order queues show it);
* ``LaneTest``: during the first replays of a new graph -- real cycles of the fit -- one lane per replay is kept busy by a
  one-thread spin kernel of 1.5 ms (no compute unit worth mentioning, but its whole hardware queue -- and longer than a
  cycle: the side branch has ~0.35 ms of slack under the selection kernel, a short spin hides in it): a replay that takes
  0.5 ms longer than the unspun ones has its chain or its side branch on that queue.  The scene update gets the first lane
  that neither stretches a replay nor shares the launch stream's queue.  Cost: one or two stretched cycles per graph
  (1-2 ms of a 200-ms fit), no stream created or destroyed, nothing that touches the arithmetic;
* ``LanePicker``: once the scene update runs, it spends four cycles on each lane the test left (and on the pooled torch
  stream of rounds 2-5) and keeps the one with the shortest cycles.

MHHIP_QUEUE_PLAN=0 switches it off (the scene update then runs on a pooled torch stream, as in rounds 2-5).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

_PLANS = {}
PROBE_US = 80.0
SPIN_US = 1500.0       # longer than a cycle: whatever shares the spun queue ends behind it
BUSY_MS = 0.5          # a replay this much longer than the unspun ones ran behind the spin


def enabled():
    return os.environ.get('MHHIP_QUEUE_PLAN', '1') != '0'


def _new():
    p = ctypes.c_void_p()
    _lib.check(_lib.lib().mh_stream_create(ctypes.byref(p)))
    return p.value


def _destroy(s):
    _lib.check(_lib.lib().mh_stream_destroy(s))


def shares(a, b, spin_us=PROBE_US):
    """do the raw streams a and b (0 / None = the default stream) drain through the same hardware queue?"""
    r = ctypes.c_int(0)
    _lib.check(_lib.lib().mh_streams_share_queue(a or None, b or None, float(spin_us), ctypes.byref(r)))
    return bool(r.value)


def spin(raw, us=SPIN_US):
    _lib.check(_lib.lib().mh_stream_spin(raw or None, float(us)))


class QueuePlan(object):
    def __init__(self, device):
        self.dev = torch.device(device)
        self._lanes = None        # raw streams, one per hardware queue
        self._views = {}
        self.stats = {'created': 0, 'probes': 0, 'lanes': 0, 'tests': 0, 'busy': []}

    def lanes(self, max_new=16):
        """one raw stream per hardware queue of this process (built once; the streams live as long as the process)"""
        if self._lanes is None:
            lanes, extra = [], []
            while len(extra) < 3 and self.stats['created'] < max_new:
                x = _new()
                self.stats['created'] += 1
                self.stats['probes'] += len(lanes)
                # (a stream on a queue we already have a lane on is kept until the end: destroyed now, its slot -- the same
                # queue -- would be dealt out again at once)
                (extra if any(shares(l, x) for l in lanes) else lanes).append(x)
            for x in extra:
                _destroy(x)
            self._lanes = lanes
            self.stats['lanes'] = len(lanes)
        return self._lanes

    def view(self, raw):
        v = self._views.get(raw)
        if v is None:
            v = self._views[raw] = torch.cuda.ExternalStream(raw, device=self.dev)
        return v

    def free_lanes(self, main_raw):
        """the lanes that do not share the launch stream's hardware queue"""
        out = []
        for l in self.lanes():
            self.stats['probes'] += 1
            if not shares(main_raw, l):
                out.append(l)
        return out

    def scene_stream(self, main_raw):
        """a first choice for the scene update's stream (``LaneTest`` may move it): any lane off the launch stream's queue"""
        free = self.free_lanes(main_raw)
        if not free:
            raise _lib.MhError('no hardware queue beside the launch stream\'s for the scene update (GPU_MAX_HW_QUEUES=1?)')
        return self.view(free[-1])


class LaneTest(object):
    """Which lanes does a captured graph keep busy?  ``before()`` / ``after()`` bracket every replay of the graph until
    ``done``; ``choice`` is then the raw stream the scene update should use (None: keep what it has)."""

    def __init__(self, plan, main_raw, reference=2):
        self.plan = plan
        self.main = main_raw
        self.cand = plan.free_lanes(main_raw)
        self.order = list(self.cand) + [None] * reference        # a spun replay per lane, then unspun ones to compare with
        self.evs = []
        self.done = len(self.cand) <= 1
        self.choice = self.cand[0] if len(self.cand) == 1 else None
        self.busy = []

    def before(self, stream):
        i = len(self.evs)
        if self.order[i] is not None:
            spin(self.order[i])
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        self._ev0 = ev

    def after(self, stream):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        self.evs.append((self._ev0, ev))
        self.recorded = len(self.evs) == len(self.order)

    def poll(self):
        """all replays of the test recorded: evaluate as soon as the device has passed the last one (no host stall)"""
        if self.done or not getattr(self, 'recorded', False) or not self.evs[-1][1].query():
            return self.done
        ms = [a.elapsed_time(b) for a, b in self.evs]
        ref = float(np.median(ms[len(self.cand):]))
        self.busy = [l for l, t in zip(self.cand, ms) if t > ref + BUSY_MS]
        quiet = [l for l, t in zip(self.cand, ms) if t <= ref + BUSY_MS]
        self.choice = quiet[0] if quiet else None
        self.ms = ms
        self.done = True
        self.plan.stats['tests'] += 1
        self.plan.stats['busy'].append([self.cand.index(l) for l in self.busy])
        return True


class LanePicker(object):
    """Closed loop behind the lane test: the scene update runs ``per`` cycles on each candidate stream, the candidate with the
    shortest cycles (median, the first cycle after a switch left out) keeps it.  The lane test says which queues the graph
    occupies; what a cycle costs beside the update is only known by running it -- at C3 the candidates that pass the test
    still differ by 0.07 ms per cycle (which other streams of the process share their queues, and with what, cannot be seen).
    ``tick(main)`` is called at the start of every cycle that launches a scene update; it returns the stream that update
    should use, or None for "stay".  Nothing waits on the host: the marks are read once the device has passed them."""

    def __init__(self, cands, per=4):
        self.cands = list(cands)
        self.per = int(per)
        self.marks = []
        self.done = len(self.cands) <= 1
        self.best = None
        self.ms = None

    def tick(self, stream):
        if self.done:
            return None
        total = self.per * len(self.cands)
        n = len(self.marks)
        if n <= total:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream)
            self.marks.append(ev)
        if n < total:
            return self.cands[n // self.per] if n % self.per == 0 else None
        if not self.marks[-1].query():
            return None
        self.ms = []
        for i in range(len(self.cands)):
            d = [self.marks[j].elapsed_time(self.marks[j + 1]) for j in range(i * self.per + 1, (i + 1) * self.per)]
            self.ms.append(float(np.median(d)))
        self.best = int(np.argmin(self.ms))
        self.done = True
        return self.cands[self.best]


def plan(device):
    key = torch.device(device).index or 0
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = QueuePlan(device)
    return p
