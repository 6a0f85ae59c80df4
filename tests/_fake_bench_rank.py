"""Stand-in rank program for the supervisor tests of bench.py (VC_BENCH_TEST_CHILD): no torch, no GPU.
FAKE_MODE: "stall_first"  -- attempt without VC_SP_TRANSPORT marks "started" and then sleeps instead of bringing communicators up;
                             with VC_SP_TRANSPORT=torch it marks "started", "up", prints one JSON line on rank 0 and exits 0
           "stall_always" -- every attempt sleeps after "started"
           "die_in_bringup" -- rank 1 exits 3 between "started" and "up" when no transport is forced; fine with torch
           "ok"           -- marks both, prints, exits 0"""
import json
import os
import sys
import time

d, r, a = os.environ["VC_BENCH_STATUS_DIR"], os.environ["RANK"], os.environ.get("VC_BENCH_ATTEMPT", "0")
mode = os.environ.get("FAKE_MODE", "ok")
forced = os.environ.get("VC_SP_TRANSPORT")


def mark(name):
    open(os.path.join(d, f"rank{r}.a{a}.{name}"), "w").close()


print(f"fake rank {r} attempt {a} transport {forced}", file=sys.stderr, flush=True)
mark("started")
if mode == "stall_always" or (mode == "stall_first" and forced is None):
    print(f"fake rank {r}: blocking in the rendezvous", file=sys.stderr, flush=True)
    time.sleep(600)
if mode == "die_in_bringup" and forced is None and r == "1":
    time.sleep(0.5)
    print("fake rank 1: ncclCommInitRank failed", file=sys.stderr, flush=True)
    sys.exit(3)
if mode == "die_in_bringup" and forced is None:
    time.sleep(600)
mark("up")
time.sleep(0.3)
if r == "0":
    print(json.dumps({"value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"]), "transport": forced or "default",
                      "master_port": os.environ["MASTER_PORT"]}), flush=True)
