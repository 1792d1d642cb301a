#!/usr/bin/env python3
"""Busy-loop on N processes for S seconds: the CPU-contention experiment of DESIGN.md (a shared host whose other tenants
saturate the cores while bench.py runs).  usage: cpu_hog.py N S"""
import multiprocessing
import sys
import time


def burn(seconds):
    end = time.time() + seconds
    x = 0
    while time.time() < end:
        for _ in range(100000):
            x = (x * 1103515245 + 12345) & 0xFFFFFFFF


if __name__ == "__main__":
    n, s = int(sys.argv[1]), float(sys.argv[2])
    ps = [multiprocessing.Process(target=burn, args=(s,)) for _ in range(n)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
