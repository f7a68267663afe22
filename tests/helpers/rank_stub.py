"""A bench.py rank without the GPU: started by bench.launch_ranks in tests/test_bench_launcher.py.  Does what a
rank does around the solve -- reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment, takes its shard of the
one seeded global batch, gathers to rank 0 (gloo), and rank 0 prints one JSON line."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

args = bench.parse_args(sys.argv[1:])
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
if os.environ.get("CPMPC_STUB_FAIL_RANK") is not None:   # a rank that dies early while the others would wait for it
    if rank == int(os.environ["CPMPC_STUB_FAIL_RANK"]):
        for i in range(60):   # more than the launcher relays: the LAST 40 must be the ones it prints
            sys.stderr.write("stub rank %d diagnostic line %d\n" % (rank, i))
        sys.stderr.write("stub rank %d: ncclCommInitRank failed (pretend)\n" % rank)
        sys.exit(3)
    import time
    time.sleep(120)
    sys.exit(0)
assert world == args.gpus and os.environ["CPMPC_BENCH_SPAWNED"] == "1"
dist.init_process_group("gloo", rank=rank, world_size=world)
sharding = importlib.import_module("cart-pole-mpc_amd.sharding")
total = world * args.batch
lo, hi = sharding.shard_range(total, rank, world)
x = bench.synth_states(bench.SEED, total, lo, hi)
g = sharding.ResultGather(4, hi - lo, torch.float64, "cpu", dst=0, depth=2)
slot = g.submit(torch.tensor(x))
g.wait_slot(slot)
locals_ = [None] * world
dist.all_gather_object(locals_, int(os.environ["LOCAL_RANK"]))
t = sharding.max_over_ranks(float(rank), "cpu")
per_rank = sharding.all_ranks([float(rank), 10.0 * rank], "cpu")
if rank == 0:
    full = g.assembled(slot).numpy()
    print("noise before the line")
    print(json.dumps({"n_gpus": world, "world_size_seen": dist.get_world_size(), "gathered": list(full.shape),
                      "in_global_order": bool(np.array_equal(full, bench.synth_states(bench.SEED, total))),
                      "local_ranks": locals_, "max_rank": t, "per_rank": per_rank}))
dist.barrier()
dist.destroy_process_group()
