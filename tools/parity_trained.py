"""A weight family that has LEFT the initial regime (/root/reference/train.py:195-225), and the parity distribution on it (GPU box).

    python tools/parity_trained.py [--steps 5000] [--pairs 128]

Trains this repo's GMMReg from PyTorch's default initialisation with train.py's recipe -- Adam(1e-4, weight_decay 1e-4), MultiStepLR(milestones = 75 / 150 /
200 "epochs", gamma 0.1: an epoch is `steps / 250` steps here), the reference's crop sample chain made on the device (32 pairs of 717 points per step from a
pool of 256 synthetic shapes) -- then reports the regime the weights are in (attention sharpness per transformer, overlap-score range, spread of the
BatchNorm statistics: from the CPU oracle on two pairs) and runs EVERY pair of configs[1] (pairs 0..N-1) and of the N = 717 / J = 16 training shape (pairs
300..300+N-1) through the HIP forward against the oracle with the same weights, with the tail pairs characterised as tests/parity_util.py does."""
import argparse
import os
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ogmm_oracle as O  # noqa: E402
from ogmm_amd import augment, synth  # noqa: E402
from ogmm_amd.gmmreg import GMMReg  # noqa: E402
from ogmm_amd.trainer import Trainer  # noqa: E402
from parity_util import ILL_CONDITIONED, distribution, reference_spread  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    model = GMMReg(512, 16, cfg).to(dev)
    tr = Trainer(model, lr=1e-4, graph=True)
    epoch = max(1, args.steps // 250)
    sched = torch.optim.lr_scheduler.MultiStepLR(tr.optimizer, milestones=[75 * epoch, 150 * epoch, 200 * epoch], gamma=0.1)          # train.py:200-202
    pool = torch.stack([torch.from_numpy(synth._patch_cloud(np.random.Generator(np.random.PCG64(500 + i)), 1024)).float() for i in range(256)]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    B = args.batch
    t0 = time.perf_counter()
    first = None
    for it in range(args.steps):
        shapes = pool[torch.randint(0, pool.shape[0], (B,), generator=gen, device=dev)]
        smp = augment.crop_pipeline(shapes, augment.draw(B, 1024, 717, gen, dev), n_out=717)
        info = tr.step(smp["src_xyz"].transpose(1, 2).contiguous(), smp["tgt_xyz"].transpose(1, 2).contiguous(), smp["transform_gt"], smp["src_overlap"], smp["tgt_overlap"])
        sched.step()
        if it == 0:
            first = float(info["loss"])
        if it % max(1, args.steps // 20) == 0 or it == args.steps - 1:
            p = {k: float(v) for k, v in info["parts"].items()}
            print("step %5d lr %.0e loss %.4f (dcp %.4f clu %.4f mse %.4f welsch %.3f) r_err %.2f deg t_err %.3f skipped %d" % (
                it, sched.get_last_lr()[0], float(info["loss"]), p["dcp"], p["clu"], p["mse"], p["welsch"], float(info["r_err_deg"]), float(info["t_err"]), tr.skipped_steps), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("# %d steps of %d pairs in %.1f s (%.0f pairs/s incl. device-side sample synthesis); loss %.3f -> %.3f; skipped steps %d" % (
        args.steps, B, dt, args.steps * B / dt, first, float(info["loss"]), tr.skipped_steps))
    model.eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # ---- the regime
    src, tgt, _, _ = synth.make_batch(0, 2, 1024, "partial")
    cap = {}
    with torch.no_grad():
        ref = O.forward(P, cfg, src, tgt, synth.fps_starts_for(0, 2, 1024), cap)
    bn = [k[:-len(".running_var")] for k in P if k.endswith(".running_var")]
    rv = torch.cat([P[k + ".running_var"].flatten() for k in bn])
    rm = torch.cat([P[k + ".running_mean"].flatten() for k in bn])
    gw = torch.cat([P[k + ".weight"].flatten() for k in bn])
    print("# regime: attention mean max-probability %s (uniform = 0.0078; default fill 0.008, sharp fill 0.27-0.47); overlap scores %.4f ... %.4f (1st / 99th percentile %.3f / %.3f)" % (
        ["%.3f" % cap["attn_maxprob_" + t_] for t_ in ("sattn1", "cattn", "sattn2")], float(min(ref[2].min(), ref[3].min())), float(max(ref[2].max(), ref[3].max())),
        float(torch.quantile(torch.cat([ref[2].flatten(), ref[3].flatten()]), 0.01)), float(torch.quantile(torch.cat([ref[2].flatten(), ref[3].flatten()]), 0.99))))
    print("# regime: BatchNorm running_var %.3g ... %.3g (median %.3g), running_mean %.3g ... %.3g, weight %.3g ... %.3g over %d channels" % (
        float(rv.min()), float(rv.max()), float(rv.median()), float(rm.min()), float(rm.max()), float(gw.min()), float(gw.max()), rv.numel()))
    # ---- every pair against the oracle, the tail characterised
    for label, N, first_pair in (("cfg1 (N=1024 J=16)", 1024, 0), ("training shape (N=717 J=16)", 717, 300)):
        r, t, o, (s_, t_, st_) = distribution(model, P, cfg, first_pair, args.pairs, N, "partial", label="trained weights (%d steps), %s pairs %d..%d" % (args.steps, label, first_pair, first_pair + args.pairs - 1))
        assert not model.fp16_overflowed()
        bad = [int(i) for i in torch.nonzero((r >= 1e-5) | (t >= 1e-5)).flatten()]
        print("# %s: %d of %d pairs within 1e-5; R max %.2e median %.2e p90 %.2e; t max %.2e; overlap-score max %.2e" % (
            label, args.pairs - len(bad), args.pairs, r.max(), r.median(), torch.quantile(r.double(), 0.9), t.max(), o.max()))
        for i in bad:
            sr, stt, probes = reference_spread(P, cfg, s_[i:i + 1], t_[i:i + 1], st_[:, i:i + 1])
            print("#   pair %d: HIP R %.2e t %.2e | reference's own spread R %.2e t %.2e (ill-conditioned: %s; HIP / spread %.2f) [%s]" % (
                first_pair + i, r[i], t[i], sr, stt, sr >= ILL_CONDITIONED, r[i] / max(sr, 1e-12), " ".join("%s %.1e" % kv for kv in probes.items())))
    return 0


if __name__ == "__main__":
    sys.exit(main())
