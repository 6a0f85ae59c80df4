"""Stand-in rank program for the supervisor tests of bench.py (VC_BENCH_TEST_CHILD): no torch, no GPU.
FAKE_MODE: "stall_first"  -- attempt without VC_SP_TRANSPORT marks "started" and then sleeps instead of bringing communicators up;
                             with VC_SP_TRANSPORT=torch it marks "started", "up", prints one JSON line on rank 0 and exits 0
           "stall_always" -- every attempt sleeps after "started"
           "die_in_bringup" -- rank 1 exits 3 between "started" and "up" when no transport is forced; fine with torch
           "hang_in_alt"  -- marks both, rank 0 SAVES the first layout's line (rank0.a<attempt>.partial) and then every rank blocks, as a rank
                             would inside an alternative layout's collective
           "die_in_alt"   -- the same, but rank 1 exits 5 after the line was saved
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
if mode in ("hang_in_alt", "die_in_alt"):
    if r == "0":
        with open(os.path.join(d, f"rank0.a{a}.partial"), "w") as fh:
            fh.write(json.dumps({"value": 2.5, "n_gpus": int(os.environ["WORLD_SIZE"]), "alt_note": "first layout only"}) + "\n")
    time.sleep(0.5)
    if mode == "die_in_alt" and r == "1":
        print("fake rank 1: the ring pass failed", file=sys.stderr, flush=True)
        sys.exit(5)
    time.sleep(600)
if r == "0":
    print(json.dumps({"value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"]), "transport": forced or "default",
                      "master_port": os.environ["MASTER_PORT"]}), flush=True)
