"""CPU, world_size 2, gloo: the multi-GPU training logic of ogmm_amd/trainer.py -- per-rank shard loss, one flat SUM
all-reduce of the gradients, rank-0 BatchNorm buffers broadcast -- with the graph running on the plain-PyTorch operation
set (tests/train_ref.py).  Expected values come from ONE process that evaluates both shards in turn."""
import os
import socket
import sys
from argparse import Namespace

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, J, TOPK = 192, 8, 96
CFG = dict(gnn_k=12, num_heads=4, km_clusters=32, overlap_radius=0.035)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(rank_pairs):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ogmm_amd import synth
    from ogmm_amd.gmmreg import GMMReg
    from train_ref import RefTrainOps
    model = GMMReg(512, J, Namespace(**CFG))
    synth.fill_state_dict(model.state_dict())
    model._train_ops = RefTrainOps()
    batch = synth.make_train_batch(rank_pairs[0], rank_pairs[1] - rank_pairs[0], N)
    starts = synth.fps_starts_for(rank_pairs[0], rank_pairs[1] - rank_pairs[0], N)
    return model, batch, starts


def _worker(rank, world, port, out_dir):
    torch.set_num_threads(3)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from ogmm_amd import dist as odist
    from ogmm_amd.trainer import Trainer
    dist = odist.init("gloo", rank, world)
    lo, hi = odist.shard_pairs(rank, world, 1, first_pair=900)
    model, batch, starts = _make((lo, hi))
    tr = Trainer(model, lr=1e-3, welsch_top_k=TOPK, dist=dist, world=world)
    info = tr.step(*batch, fps_starts=starts)
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    torch.save({"grads": grads, "loss": info["loss"], "state": {k: v.clone() for k, v in model.state_dict().items()}},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_training_step(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    # both ranks end the step with identical gradients, parameters and BatchNorm buffers
    for k in got[0]["grads"]:
        assert torch.equal(got[0]["grads"][k], got[1]["grads"][k]), k
    for k in got[0]["state"]:
        assert torch.equal(got[0]["state"][k], got[1]["state"][k]), k

    # expected: each shard evaluated by one process, gradients of (rest_r / W + clu_r) summed; buffers of shard 0
    torch.set_num_threads(6)
    sys.path.insert(0, ROOT)
    from ogmm_amd.trainer import Trainer
    total, buffers0, losses_ = None, None, []
    for r in range(world):
        model, batch, starts = _make((900 + r, 901 + r))
        tr = Trainer(model, lr=1e-3, welsch_top_k=TOPK, dist=None, world=1)
        tr.world = world                                   # the loss split of a 2-rank job, without the collectives
        model.train()
        out = model(batch[0], batch[1], fps_starts=starts)
        loss, _ = tr.local_loss(out, *batch)
        loss.backward()
        losses_.append(loss.detach())
        g = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        total = g if total is None else {k: total[k] + g[k] for k in g}
        if r == 0:
            buffers0 = {k: v.clone() for k, v in model.named_buffers()}
    for r in range(world):
        assert torch.allclose(got[r]["loss"], losses_[r], rtol=1e-5)
    gnorm = torch.sqrt(sum((v.double() ** 2).sum() for v in total.values()))
    for k, v in total.items():
        err = (got[0]["grads"][k].double() - v.double()).norm()
        assert err <= 2e-3 * v.double().norm() + 1e-6 * gnorm, (k, float(err), float(v.norm()))
    for k, v in buffers0.items():
        assert torch.allclose(got[1]["state"][k].double(), v.double(), rtol=1e-5, atol=1e-6), k
    # and the parameters moved (Adam step applied after the all-reduce)
    fresh, _, _ = _make((900, 901))
    moved = sum(float((got[0]["state"][k] - v).abs().sum()) for k, v in fresh.state_dict().items() if v.is_floating_point() and "running" not in k)
    assert moved > 0


def _bench_train_worker(rank, world, port, out_dir):
    """bench.py's train leg (`--workload train`) under torch.distributed (gloo, CPU) through its OGMM_BENCH_STUB seam: the code `python bench.py --gpus 8
    --workload train` runs around the model, with the training graph on tests/train_ref.py's operation set at a toy size"""
    import contextlib
    import io
    torch.set_num_threads(3)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), OGMM_BENCH_STUB="1")
    sys.argv = ["bench.py", "--gpus", str(world), "--workload", "train", "--steps", "2", "--warmup", "1"]
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(os.path.join(out_dir, "train_out%d.txt" % rank), "w") as f:
        f.write(buf.getvalue())


def test_bench_train_leg_under_two_ranks(tmp_path):
    """`bench.train_leg` had never run at world > 1.  Two ranks, gloo: one JSON line from rank 0, the whole-job value (pairs of
    both ranks / the slowest rank's time), a data-parallel label, and both ranks end with identical parameters and buffers."""
    import json
    world = 2
    mp.spawn(_bench_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out0 = open(os.path.join(str(tmp_path), "train_out0.txt")).read().strip().splitlines()
    out1 = open(os.path.join(str(tmp_path), "train_out1.txt")).read().strip()
    assert len(out0) == 1 and out1 == "", "exactly ONE JSON line, printed by rank 0"
    line = json.loads(out0[0])
    assert line["metric"] == "train_pairs_per_sec" and line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert abs(line["value"] - 2 * 1 * 2 / (line["ms_per_step"] * 2 / 1e3)) < 1e-6 * line["value"]          # pairs of ALL ranks / the max-over-ranks time
    assert "data parallel x2" in line["config"]["parallelism"]
    assert line["stub_ranks_agree"] is True and line["final_loss"] == line["final_loss"] and line["final_loss"] > 0
