#!/usr/bin/env python3
"""Where the Init phase of `jellyfish-amd count` goes at the metric's geometry: table creation (allocation + clearing),
workspace reservation, the feed's pinned buffers.  usage: python tools/init_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
from jellyfish_amd import capi
capi.load()
print("load lib %.2f s" % (time.time() - t0), flush=True)
t0 = time.time(); n = capi.device_count(); print("device_count %.2f s" % (time.time() - t0), flush=True)
for rep in range(2):
    t0 = time.time()
    t = capi.Table(21, 1 << 34, canonical=True)
    t1 = time.time()
    t.reserve(10 << 30)
    t.wait()
    t2 = time.time()
    p = capi.Parser(21)
    t3 = time.time()
    print("rep %d: create table %.2f s, reserve workspace %.2f s, parser %.2f s" % (rep, t1 - t0, t2 - t1, t3 - t2), flush=True)
    p.close(); t.close()
# a table created right after a big one was freed waits for the driver to take the freed memory back (scrubbing): how long?
for pause in (0.0, 2.0, 6.0):
    t = capi.Table(21, 1 << 34, canonical=True); t.reserve(10 << 30); t.wait(); t.close()
    time.sleep(pause)
    t1 = time.time(); t = capi.Table(21, 1 << 34, canonical=True); t2 = time.time(); t.close()
    print("pause %.0f s after freeing table + workspace: create %.2f s" % (pause, t2 - t1), flush=True)
