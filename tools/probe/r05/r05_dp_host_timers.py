#!/usr/bin/env python
"""Host time the data-parallel reducer spends per step, by method (bench.py --workload c3 --force-dp; the backward runs on autograd's
thread, which cProfile does not see, so the methods are wrapped with wall-clock timers)."""
import atexit, os, runpy, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from neko_amd import dp as _dp
from neko_amd import engine as _engine

acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or f"{getattr(obj, '__name__', obj)}.{name}"

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label]
            e[0] += 1
            e[1] += time.perf_counter() - t
    setattr(obj, name, w)


for m in ("group_ready", "_reduce_range", "flush", "finish", "reduce_flags"):
    wrap(_dp.GradReducer, m, "GradReducer." + m)
wrap(dist, "all_reduce", "dist.all_reduce")
wrap(_engine.SideStream, "join", "SideStream.join")
steps = int(sys.argv[sys.argv.index("--steps") + 1]) + 10 if "--steps" in sys.argv else 60


@atexit.register
def report():
    for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:28s} calls/step {n / steps:6.1f}   ms/step {1e3 * t / steps:7.3f}   us/call {1e6 * t / max(n, 1):7.1f}", file=sys.stderr)


sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"), run_name="__main__")
