#!/usr/bin/env python3
"""What does a sum all-reduce cost inside a ONE-rank RCCL group?  HIP-event time of dist.all_reduce on device tensors of several sizes
(in place, as distributed.OverlappedGradReducer issues it), synchronous and async_op=True."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for mb in (0.06, 1, 8, 16, 72):
    t = torch.ones(int(mb * 1e6 / 4), device=dev)
    for _ in range(3):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dist.all_reduce(t)
    e1.record()
    torch.cuda.synchronize()
    sync_us = e0.elapsed_time(e1) / 20 * 1e3
    e0.record()
    works = [dist.all_reduce(t, async_op=True) for _ in range(20)]
    works[-1].wait()
    e1.record()
    torch.cuda.synchronize()
    print(f"{mb:6.2f} MB: all_reduce {sync_us:8.1f} us per call ({mb * 1e6 / sync_us / 1e3:7.1f} GB/s if it moved the data once)   async {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us   value {float(t[0])}")
dist.destroy_process_group()
