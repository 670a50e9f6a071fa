"""GPU: data-parallel training = single process at the same global batch.
Two ranks share the one GPU of the test box over gloo (RCCL refuses two ranks on one
device; the collective semantics under test — SUM all-reduce of the flat gradient, no
averaging, identical Adam step — are backend-independent), and bench.py's multi-rank
control flow is exercised the same way."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK_SCRIPT = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from cellulus_amd import parallel
from cellulus_amd.criterions import get_loss
from cellulus_amd.models import get_model
from cellulus_amd.optim import Adam
from cellulus_amd.train import train_iteration
rank, world, local = parallel.init_from_env()
dev = torch.device("cuda", local)
cfg = dict(in_channels=1, out_channels=2, num_fmaps=8, fmap_inc_factor=3, features_in_last_layer=16,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
torch.manual_seed(100 + rank)                       # different initial weights per rank ...
model = get_model(**cfg).to(dev)
flat, _ = model.flatten_parameters()
parallel.broadcast_(flat, src=0)                    # ... made identical by the broadcast
crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=dev)
opt = Adam(model.parameters(), lr=1e-3, weight_decay=0.01)
data = np.load(sys.argv[2])
per = data["raw0"].shape[0] // world                # crops per rank
losses = []
for step in range(2):
    raw = torch.from_numpy(data[f"raw{step}"][rank * per:(rank + 1) * per])
    a = torch.from_numpy(data[f"a{step}"][rank * per:(rank + 1) * per])
    r = torch.from_numpy(data[f"r{step}"][rank * per:(rank + 1) * per])
    loss, _, _ = train_iteration((raw, a, r), model, crit, opt, dev)
    losses.append(loss)
    if step == 0:
        grad0 = model._flat_grad.clone()            # the all-reduced gradient of the first step (same weights everywhere)
from cellulus_amd.models.plan import DualPlan
assert isinstance(next(iter(model._plans.values())), DualPlan) == (os.environ["CLX_STREAMS_MIN_GFLOP"] == "0")
# every rank issued the same gradient ranges in the same order (they depend on the plan only) ...
ranges = [None] * world
torch.distributed.all_gather_object(ranges, model._last_bucket_ranges)
assert all(r == ranges[0] for r in ranges), ranges
assert ranges[0][-1][0] == 0 and ranges[0][0][1] == model._flat_grad.numel()
assert all(a[0] == b[1] for a, b in zip(ranges[0], ranges[0][1:]))        # a partition of the flat buffer
# ... and ends with the same parameters
mine = model._flat.clone()
parallel.broadcast_(mine, src=0)
assert torch.equal(mine, model._flat), "ranks diverged"
if rank == 0:
    np.savez(sys.argv[3], flat=model._flat.cpu().numpy(), losses=np.array(losses), nbuckets=len(ranges[0]),
             grad0=grad0.cpu().numpy())
torch.distributed.barrier()
"""


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(script, args, env_extra, nproc=2, port=None):
    port = port or _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc),
               CLX_DIST_BACKEND="gloo", CLX_LOCAL_DEVICE="0", **env_extra)
    procs = [subprocess.Popen([sys.executable, script, *args], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nproc)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return outs


@pytest.mark.parametrize("world,bucket_mb,dual", [(2, "4", False), (2, "0.002", False), (2, "0", False), (8, "4", False),
                                                  (8, "0.002", False), (2, "0.002", True), (2, "0", True)])
def test_ranks_equal_one_process_at_global_batch(tmp_path, device, world, bucket_mb, dual):
    """dual: every rank runs its crops as two halves on two streams (DualPlan, forced at this size); the buckets
    must leave in the same order.  bucket_mb: default (one bucket at this model size), 2 KB buckets (every layer goes out on its own
    while the backward pass continues), 0 (single all-reduce after the backward pass).  world 8 = the
    node size of BASELINE configs[2] (one crop per rank here): same bucket sequence on every rank, same
    parameters on every rank, and those of one process at the global batch."""
    from cellulus_amd.criterions import get_loss
    from cellulus_amd.models import get_model
    from cellulus_amd.optim import Adam
    from cellulus_amd.train import train_iteration

    rng = np.random.default_rng(0)
    data = {}
    G = 4 if world == 2 else 8                             # global batch
    for step in range(2):
        data[f"raw{step}"] = rng.random((G, 1, 44, 52)).astype(np.float32)
        a = np.repeat(rng.integers(3, 25, size=(G, 40, 2)), 5, axis=1)
        data[f"a{step}"] = a.astype(np.int64)
        data[f"r{step}"] = (a + rng.integers(1, 3, size=a.shape)).astype(np.int64)
    np.savez(tmp_path / "data.npz", **data)
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    _launch(str(script), [ROOT, str(tmp_path / "data.npz"), str(tmp_path / "out.npz")],
            {"CLX_GRAD_BUCKET_MB": bucket_mb, "CLX_STREAMS_MIN_GFLOP": "0" if dual else "1e9"}, nproc=world)
    got = np.load(tmp_path / "out.npz")
    assert got["nbuckets"] == 1 if bucket_mb != "0.002" else got["nbuckets"] > 4

    # single process at the global batch, same initial weights (rank 0's seed)
    cfg = dict(in_channels=1, out_channels=2, num_fmaps=8, fmap_inc_factor=3, features_in_last_layer=16,
               downsampling_factors=[[2, 2]], num_spatial_dims=2)
    torch.manual_seed(100)
    model = get_model(**cfg).to(device)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=device)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=0.01)
    losses = []
    for step in range(2):
        batch = tuple(torch.from_numpy(data[f"{k}{step}"]) for k in ("raw", "a", "r"))
        loss, _, _ = train_iteration(batch, model, crit, opt, device)
        losses.append(loss)
        if step == 0:
            grad0 = model._flat_grad.cpu().numpy().copy()
    # the loss is a SUM over pairs: the all-reduced loss equals the global-batch loss
    np.testing.assert_allclose(got["losses"], losses, rtol=1e-5)
    # the all-reduced gradient of the first step IS the global batch's gradient (sums of the same terms in another order)
    scale = np.abs(grad0).max()
    # (terms of 1e3 cancel in a weight's gradient, so the order of the additions is worth ~1e-4 of the largest element on a
    # few of them: round 6 saw 4 of 16762 at 1.3e-4 with eight ranks once in some forty runs.  Nearly all within 1e-4,
    # none further than 5e-4)
    gdiff = np.abs(got["grad0"] - grad0) / scale
    assert (gdiff > 1e-4).mean() < 1e-3 and gdiff.max() < 5e-4, ((gdiff > 1e-4).sum(), gdiff.max())
    # ... and two Adam steps later the parameters are the one-process parameters.  Adam's first steps move a parameter by
    # lr * g / (|g| + eps) ~ +-lr whatever |g| is, so where a gradient element is smaller than the float noise of its
    # sum (terms of 1e3 cancelling) the ORDER of the additions decides its sign — atomics: it changes from run to run,
    # about one run in twenty lands 1 % of the elements on the other side (|difference| = 2.8 lr, evidence kept below).
    # The bar: nearly all elements to 2e-5, none further than the steps can move them apart.
    ref_flat = model._flat.cpu().numpy()
    far = np.abs(got["flat"] - ref_flat) > 2e-5
    if far.any():                                                  # keep the evidence
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", f"ddp_mismatch_w{world}_b{bucket_mb}_{int(dual)}.npz"),
                 ranks=got["flat"], one=ref_flat, grad_ranks=got["grad0"], grad_one=grad0)
    assert far.mean() < 0.03, far.mean()
    assert np.abs(got["flat"] - ref_flat).max() < 2 * 2 * 1e-3 * 1.05


def test_bench_starts_eight_ranks(tmp_path):
    """`python bench.py --gpus 8` as the driver's scaling run issues it (no rank environment): eight
    fresh children, one JSON line from rank 0 with what the backend saw and what went over the wire."""
    env = dict(os.environ, CLX_DIST_BACKEND="gloo", CLX_LOCAL_DEVICE="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--workload", "tiny", "--no-infer"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and len(rec["per_rank_ms_per_step"]) == 8
    assert rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp8" and rec["scaling"] == "weak"
    # the gradient exchange: the whole flat f32 gradient (+ the four float64 loss sums), as buckets
    assert sum(rec["allreduce_bucket_bytes"]) + 32 == rec["allreduce_bytes"] and rec["allreduce_bytes"] > 4 * 100000
    assert rec["loader_procs"] == 0 and rec["host_cores_per_rank"] >= 1


def test_cfg3_dress_rehearsal_eight_ranks_at_the_real_workload(tmp_path, device):
    """BASELINE configs[2] (the cfg-2 network, batch 8 per rank, 8 ranks) at its REAL workload: `python bench.py --gpus 8` as
    a scaling driver would issue it, the eight ranks sharing the one GPU of this box over gloo (8 x ~8 GB fit 288 GB; RCCL
    refuses several ranks per device — the collective semantics, the bucket ranges of the 38.5-MB gradient, the loader
    policy and the control flow are what is under test, not the wire).  The line must carry both halves of the metric
    (train crops/s, infer Mpixels/s through the sharded infer()) and the real train() with per-rank loader processes."""
    import gc

    import bench

    # the eight ranks need the device's memory: give back what this process's earlier tests left in torch's allocator
    gc.collect()
    torch.cuda.empty_cache()
    env = dict(os.environ, CLX_DIST_BACKEND="gloo", CLX_LOCAL_DEVICE="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "CLX_STREAMS",
              "CLX_GRAD_BUCKET_MB", "CLX_DEVICE_PAIRS"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--no-train3d", "--infer-samples", "2", "--e2e-iterations", "18"]
    p = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=3000)
    if p.returncode != 0:
        # eight ranks over gloo's TCP transport on loopback, beside their loader processes: twice in round 6 a rank lost its
        # connection ("Connection closed by peer").  The wire is not what is under test: keep the evidence, and run the
        # command ONCE more if — and only if — the failure is the transport's
        import re

        tail = (p.stdout + p.stderr)[-6000:]
        free_b, total_b = torch.cuda.mem_get_info()
        tail += (f"\n[parent process] device free {free_b / 2 ** 30:.1f} of {total_b / 2 ** 30:.1f} GiB after the run; its own tensors "
                 f"{torch.cuda.memory_allocated() / 2 ** 30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2 ** 30:.2f} GiB\n")
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            with open(os.path.join(ROOT, "gpurun_out", "cfg3_dress_rehearsal_first_failure.txt"), "w") as fh:
                fh.write(tail)
        if re.search(r"Connection (closed|reset)|Broken pipe|Socket Timeout|gloo.*(timeout|timed out)|EADDRINUSE|address already in use",
                     tail, re.I):
            p = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=3000)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    lines = [l for l in p.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    (tmp_path / "line.json").write_text(lines[0])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):                     # kept as evidence next to the test log
        with open(os.path.join(out_dir, "cfg3_dress_rehearsal_line.json"), "w") as fh:
            fh.write(lines[0] + "\n")
    # ---- the data-parallel step
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and rec["backend"] == "gloo" and rec["scaling"] == "weak"
    assert rec["config"]["workload"] == bench.WORKLOADS["train2d"]["name"]
    assert rec["config"]["global_batch"] == 64 and rec["config"]["parallelism"] == "dp8"
    n_params = 9624258                                           # cfg-2
    assert rec["allreduce_bytes"] == n_params * 4 + 32 == 38497064
    assert sum(rec["allreduce_bucket_bytes"]) == n_params * 4 and len(rec["allreduce_bucket_bytes"]) > 1
    assert min(rec["allreduce_bucket_bytes"][:-1]) >= rec["grad_bucket_mb"] * 2 ** 20
    assert rec["bucket_ranges_identical_on_all_ranks"] and rec["params_identical_on_all_ranks"]
    assert len(rec["per_rank_peak_mem_gb"]) == 8 and max(rec["per_rank_peak_mem_gb"]) < 288 / 8
    assert len(rec["per_rank_ms_per_step"]) == 8 and "allreduce_ms_exposed" in rec and rec["value"] > 0
    losses = rec["losses_from_first_step"]
    assert len(losses) == 3 and all(np.isfinite(losses))
    # ---- both halves of the metric, and the real train(), at N > 1
    inf = rec["infer"]
    assert inf["n_gpus"] == 8 and inf["value"] > 0 and inf["e2e"]["samples"] == 16 and inf["e2e"]["ranks"] == 8
    assert inf["at_256"]["mpixels_s"] > 0 and inf["at_256"]["tile"] == 256 and inf["e2e"]["objects_per_sample"] >= 0
    e2e = rec["train_e2e"]
    assert e2e["ranks"] == 8 and e2e["value"] > 0 and e2e["loader_procs"] >= 1
    assert e2e["pair_sampler"].startswith("device") and e2e["elastic_deform"]
    # ---- the loss is a SUM over pairs: the all-reduced loss of step 0 (seeded initial weights) is the sum of the
    # eight ranks' one-process losses on their own crops
    from cellulus_amd.train import train_iteration

    solo = []
    for r in range(8):
        model, crit, opt, batch = bench.build_step_inputs("train2d", r, device, broadcast=False)
        solo.append(train_iteration(batch, model, crit, opt, device)[0])
        del model, crit, opt, batch
        torch.cuda.empty_cache()
    assert abs(sum(solo) - losses[0]) <= 1e-5 * abs(sum(solo)), (sum(solo), losses[0], solo)


def test_bench_multi_rank_control_flow(tmp_path):
    cmd_env = {}
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2",
               CLX_DIST_BACKEND="gloo", CLX_LOCAL_DEVICE="0", **cmd_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                               "--warmup", "1", "--workload", "tiny", "--no-infer"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = [l for l in outs[0][0].strip().split("\n") if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["global_batch"] == 4
    assert rec["value"] > 0 and "roofline" in rec and "cpu_baseline" not in rec
    assert not [l for l in outs[1][0].split("\n") if l.startswith("{")]     # only rank 0 prints the JSON line
    assert rec["ranks_seen"] == 2 and len(rec["per_rank_ms_per_step"]) == 2 and "allreduce_ms_exposed" in rec


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO rank environment (the driver's scaling command): bench.py
    itself starts one fresh child per rank before touching the GPU, relays rank 0's line and
    reports how many ranks the collective backend saw.  (gloo: two ranks share the one GPU here.)"""
    env = dict(os.environ, CLX_DIST_BACKEND="gloo", CLX_LOCAL_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "tiny", "--no-infer"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["backend"] == "gloo"
    assert rec["config"]["global_batch"] == 4 and rec["value"] > 0
    # a failing rank makes the launcher exit non-zero instead of hanging
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "tiny", "--no-infer"], env=dict(env, CLX_LOCAL_DEVICE="99"), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode != 0


_RCCL_SCRIPT = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cellulus_amd import parallel
from cellulus_amd.criterions import get_loss
from cellulus_amd.models import get_model
from cellulus_amd.optim import Adam
from cellulus_amd.train import _fused_step
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1)          # "nccl" is RCCL on ROCm
cfg = dict(in_channels=1, out_channels=2, num_fmaps=8, fmap_inc_factor=3, features_in_last_layer=16,
           downsampling_factors=[[2, 2]], num_spatial_dims=2)
rng = np.random.default_rng(0)
raw = torch.from_numpy(rng.random((2, 1, 44, 52)).astype(np.float32)).to(dev)
a = np.repeat(rng.integers(3, 25, size=(2, 40, 2)), 5, axis=1)
anchor = torch.from_numpy(a.astype(np.int64)).to(dev)
reference = torch.from_numpy((a + rng.integers(1, 3, size=a.shape)).astype(np.int64)).to(dev)
results = []
for world_patch in (False, True):
    torch.manual_seed(7)
    model = get_model(**cfg).to(dev)
    crit = get_loss(temperature=10.0, regularizer_weight=1e-5, density=0.1, num_spatial_dims=2, device=dev)
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=0.01)
    if world_patch:                 # run the multi-rank code path (buckets on RCCL's stream) with one rank
        parallel.world_size = lambda: 2
        calls = []
        real = dist.all_reduce
        def counted(t, *args, **kw):
            calls.append((t.numel(), kw.get("async_op", False)))
            return real(t, *args, **kw)
        dist.all_reduce = counted
    losses = [_fused_step(model, crit, opt, raw, anchor, reference)[0] for _ in range(3)]
    torch.cuda.synchronize()
    results.append((losses, model._flat.clone()))
assert len(calls) >= 3 * 4 and all(async_op for _n, async_op in calls), calls[:8]
assert results[0][0] == results[1][0], (results[0][0], results[1][0])
assert torch.allclose(results[0][1], results[1][1], atol=1e-6)
dist.destroy_process_group()
print("rccl buckets ok", len(calls))
"""


def test_bucketed_all_reduce_runs_on_rccl(tmp_path, device):
    """The overlapped gradient reduction on the REAL RCCL backend (one rank: SUM = identity): three
    fused steps with 2 KB buckets — asynchronous collectives on RCCL's stream between the backward
    kernels, waited for before Adam — give the same losses and weights as the single-rank path."""
    script = tmp_path / "rccl.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, CLX_GRAD_BUCKET_MB="0.002", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CLX_DIST_BACKEND", "CLX_LOCAL_DEVICE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script), ROOT, str(_free_port())], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "rccl buckets ok" in p.stdout


_INFER_SCRIPT = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
os.chdir(sys.argv[2])
from cellulus_amd.configs import ExperimentConfig
from cellulus_amd.infer import infer
import tomli
cfg = ExperimentConfig(**tomli.load(open(sys.argv[3], "rb")))
torch.manual_seed(42)
np.random.seed(42)
infer(cfg)
"""


@pytest.mark.parametrize("fused,nproc", [("1", 2), ("0", 2), ("1", 8), ("0", 8)])
def test_multi_rank_inference_shards_samples_and_writes_one_dataset(tmp_path, device, fused, nproc):
    """infer() under two ranks (samples sharded, no collective on the data path; ADVICE r1: every rank
    used to create — i.e. replace — the prediction dataset): rank 0 creates each dataset once, both
    ranks fill their samples, nothing a faster rank wrote is lost.  The post-processing of each
    sample equals the single-process result on the embeddings that were written."""
    from cellulus_amd.utils import zarr_io
    from oracle import infer_oracle as IO
    from oracle.unet_oracle import OracleUNetModel

    container = str(tmp_path / "data.zarr")
    rng = np.random.default_rng(0)
    raw = rng.random((5, 1, 72, 80)).astype(np.float32)   # 5 samples: 3 + 2 on two ranks; of eight, three get none
    f = zarr_io.open(container)
    f["test/raw"] = raw
    f["test/raw"].attrs["axis_names"] = ["s", "c", "y", "x"]
    torch.manual_seed(0)
    oracle = OracleUNetModel(in_channels=1, out_channels=2, num_spatial_dims=2, num_fmaps=8, fmap_inc_factor=2,
                             features_in_last_layer=16, downsampling_factors=[[2, 2]])
    os.makedirs(tmp_path / "models")
    torch.save({"model_state_dict": oracle.state_dict()}, tmp_path / "models" / "best_loss.pth")
    (tmp_path / "infer.toml").write_text(f"""
object_size = 12
normalization_factor = 1.0
[model_config]
num_fmaps = 8
fmap_inc_factor = 2
features_in_last_layer = 16
downsampling_factors = [[2, 2]]
checkpoint = "models/best_loss.pth"
[inference_config]
crop_size = [56, 56]
num_infer_iterations = 2
p_salt_pepper = 0.05
reduction_probability = 0.5
min_size = 6
grow_distance = 2
shrink_distance = 3
device = "cuda:0"
[inference_config.dataset_config]
container_path = "{container}"
dataset_name = "test/raw"
[inference_config.prediction_dataset_config]
container_path = "{container}"
dataset_name = "embeddings"
[inference_config.detection_dataset_config]
container_path = "{container}"
dataset_name = "detection"
secondary_dataset_name = "embeddings"
[inference_config.segmentation_dataset_config]
container_path = "{container}"
dataset_name = "segmentation"
secondary_dataset_name = "detection"
""")
    script = tmp_path / "infer_rank.py"
    script.write_text(_INFER_SCRIPT)
    # fused = "1": predict -> detect -> segment per sample in device memory on each rank's samples;
    # "0": the reference's dataset-by-dataset order (three sharded passes over the zarr container)
    _launch(str(script), [ROOT, str(tmp_path), str(tmp_path / "infer.toml")], {"CLX_FUSED_INFER": fused},
            nproc=nproc)
    g = zarr_io.open(container, "r")
    emb, det, seg = g["embeddings"][...], g["detection"][...], g["segmentation"][...]
    assert emb.shape == (5, 3, 72, 80) and det.shape == seg.shape == (5, 1, 72, 80)
    for s in range(5):                                   # every sample was written by exactly one rank
        assert np.abs(emb[s, :2]).max() > 0 and emb[s, 2].max() > 0, f"sample {s} was lost"
        thr = IO.threshold_otsu(emb[s, -1])
        np.testing.assert_array_equal(g["binary-segmentation"][s, 0], (emb[s, -1] < thr).astype(np.uint16))
        ref_seg = IO.size_filter(IO.grow_shrink(det[s, 0].astype(np.int32), 2, 3), 6)
        np.testing.assert_array_equal(seg[s, 0], ref_seg.astype(np.uint16))
    assert g["embeddings"].attrs["axis_names"] == ["s", "c", "y", "x"]
