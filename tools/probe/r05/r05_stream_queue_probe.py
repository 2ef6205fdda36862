#!/usr/bin/env python
"""Do kernels on a torch pool stream run concurrently with kernels on the default stream?  (HIP streams share GPU_MAX_HW_QUEUES hardware
queues; two streams on one queue serialise.)  For the first 12 pool streams, after a warm-up launch on each (a stream's first launch creates
or binds its hardware queue: ~5 ms): a ~340-us spin on the default stream and a ~170-us spin on the candidate, enqueued back to back;
total = ~340 us when they overlap, ~510 us when they share a queue."""
import sys, torch
dev = torch.device("cuda", 0)
x = torch.zeros(1024, device=dev)
main = torch.cuda.current_stream(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(12)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
out = []
for i, s in enumerate(streams):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(main)
    torch.cuda._sleep(800_000)
    with torch.cuda.stream(s):
        torch.cuda._sleep(400_000)
    main.wait_stream(s)
    e1.record(main)
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3
    out.append(f"{'C' if t < 430 else 's'}{t:.0f}")
print(" ".join(out))
