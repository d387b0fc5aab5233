#!/usr/bin/env python3
"""probe: bench.run_config5 after N other engines have lived (and died) in the process"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
import sdr_server_amd as xl
torch.cuda.set_device(0)
ctx = {"xl": xl, "torch": torch, "dist": None, "rank": 0, "world": 1, "cuda": True, "lpf": xl.create_low_pass_filter, "dev_groups": [torch.from_numpy(bench.make_group(g)).cuda() for g in range(bench.NSRC_GROUPS)], "feed": "c"}
def c5(tag):
    m5 = bench.run_config5(ctx, 1024, 3, spot=False, blocks_per_step=320)
    print(tag, round(m5["us_per_block"], 2), "us/block; launches", round(m5["call_ms_avg"] * 1e3 / 8, 2), flush=True)
c5("config5 first")
n = 0
for kind in ("server1024", "server1024", "native1024", "server2048", "server4096", "oneblock", "staggered", "server128", "host", "config3", "dropin"):
    if kind.startswith("server"):
        bench.run_workload(ctx, int(kind[6:]), 5, 1, 1, "optimized", blocks_per_step=160)
    elif kind == "native1024":
        bench.run_workload(ctx, 1024, 5, 1, 1, "native", blocks_per_step=64, poly3=False)
    elif kind == "oneblock":
        bench.run_workload(ctx, 1024, 5, 1, 1, "optimized", group=1, poly3=False, blocks_per_step=160)
    elif kind == "staggered":
        bench.run_workload(ctx, 1024, 5, 1, 1, "optimized", staggered=True, poly3=False, blocks_per_step=160)
    elif kind == "host":
        bench.run_host_delivered(ctx, 1024, 5, calls=4, spot=False)
    elif kind == "config3":
        bench.run_config3_mixed(ctx, steps=10, spot=False)
    elif kind == "dropin":
        bench.run_config2_dropin(ctx, calls=50, spot=False)
    n += 1
    c5(f"config5 after {n:2d} engines (last: {kind})")
