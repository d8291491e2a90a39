"""N > 1 path on CPU: two processes (gloo) each own a contiguous shard of a global batch and a
replica of the controller; results gathered over the process group equal the single-process
result bitwise.  The compute backend here is the test-only emulation of the device code (no GPU in
this container); bench.py runs the same sharding with one MI355X per rank."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import qpc_pkg; qpc_pkg.load()
import emu
from qpcontrol_jl_amd import workloads as W
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B = 13  # deliberately not divisible by 2
ctrl, wl = W.atlas_workload(emu.factory, B, 3)
lo, hi, r = W.run_sharded(ctrl, wl, rank, world)
assert (lo, hi) == W.shard_bounds(B, rank, world)
# gather (padded) shards on every rank -- test-side only; the data path itself has no collective
nv = wl.mech.nv
pad = (B + world - 1) // world
buf = torch.zeros(pad, nv + 1, dtype=torch.float64)
buf[: hi - lo, :nv] = torch.from_numpy(r["tau"]); buf[: hi - lo, nv] = torch.from_numpy(r["status"].astype(np.float64))
out = [torch.zeros_like(buf) for _ in range(world)]
dist.all_gather(out, buf)
t = torch.tensor([float(hi - lo)], dtype=torch.float64)
dist.all_reduce(t)  # sum of shard sizes == B
if rank == 0:
    tau = np.concatenate([out[k][: W.shard_bounds(B, k, world)[1] - W.shard_bounds(B, k, world)[0], :nv].numpy() for k in range(world)])
    full = ctrl.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    assert int(t.item()) == B
    assert np.array_equal(tau, full["tau"]), np.abs(tau - full["tau"]).max()
    print("SHARDED_OK", B, world)
dist.destroy_process_group()
'''


def test_two_rank_sharding_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARDED_OK 13 2" in outs[0]


def test_shard_bounds_cover_batch():
    from qpcontrol_jl_amd import workloads as W
    for B in (1, 7, 8, 262144):
        for world in (1, 2, 3, 4, 8):
            b = [W.shard_bounds(B, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == B
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    """bench.py's multi-rank code path (torch.distributed.run launch, env parsing, barrier, max-over-ranks
    time, one JSON line from rank 0) on a single-GPU box: both ranks share cuda:0 through the
    QPC_BENCH_TEST_ONE_DEVICE hook (gloo instead of RCCL).  The value is meaningless, the plumbing is not."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QPC_BENCH_TEST_ONE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--batch-per-gpu", "2048"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4096 and d["scaling"] == "weak" and d["value"] > 0
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_in_library_leg_on_emulated_devices():
    """bench.py's second multi-GPU leg -- ONE process, qpc_finalize_devices(ctrl, N, ids), a resident shard per device, a launch per device
    and step through qpc_control_batch_device_on (what QPControlHIP.jl's `finalize!(...; devices = ...)` drives) -- on three fake devices of
    the CPU emulation"""
    import emu
    emu.build()
    sys.path.insert(0, ROOT)
    import bench
    r = bench.in_library_leg(3, 5, steps=2, warmup=1, config=3, ids=[0, 1, 2], lib_path=emu.EMU_LIB)
    assert r["n_devices"] == 3 and r["batch_per_device"] == 5 and r["solves_per_s"] > 0
    assert all(f >= 0.8 for f in r["optimal_fraction_per_device"]), r


@pytest.mark.gpu
def test_bench_without_a_launcher_two_ranks_on_one_device():
    """`python bench.py --gpus 2` with NO launcher: the process starts its two ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment), passes rank 0's line on with every rank's kernel time in it, and adds the in-library leg (one process; on this one-GPU box
    "devices": both cuda:0 through the QPC_BENCH_TEST_ONE_DEVICE hook).  The numbers mean nothing, the plumbing does."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QPC_BENCH_TEST_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch-per-gpu", "2048"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4096 and d["value"] > 0
    assert len(d["config"]["per_rank"]) == 2 and all(r["kernel_avg_ms"] > 0 for r in d["config"]["per_rank"])
    lib = d["config"]["in_library_devices"]  # (under the one-device hook the leg drives device 0 alone: a controller takes every device once)
    assert lib["n_devices"] == 1 and lib["solves_per_s"] > 0 and all(f > 0.9 for f in lib["optimal_fraction_per_device"]), lib
