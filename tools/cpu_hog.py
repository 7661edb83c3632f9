#!/usr/bin/env python3
"""Occupy every hardware thread of the host for SECONDS (default 60): the loaded-host leg of the update's eager-vs-graphs A/B
(a GPU box is one of eight tenants of a 256-thread host; the driver's bench box is not idle: VERDICT r4)."""
import multiprocessing as mp
import os
import sys
import time


def spin(t_end):
    x = 0
    while time.time() < t_end:
        for _ in range(100000):
            x += 1


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else os.cpu_count()
    t_end = time.time() + secs
    ps = [mp.Process(target=spin, args=(t_end,), daemon=True) for _ in range(n)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
