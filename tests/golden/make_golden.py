"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE (gfmei/ogmm, mounted at
/root/reference) on CPU in the build container.  The reference has no tests or golden vectors of its
own, so these outputs are what pins oracle/ogmm_oracle.py and, through it,
the HIP path.  The reference itself never travels: only inputs + outputs are stored here.

    python tests/golden/make_golden.py [name ...]      # rewrites tests/golden/<name>.npz (default: all eval fixtures)

Each fixture holds: the inputs (src, tgt), the six pinned FPS start draws (the reference's
`torch.randint` at lib/utils.py:190 is patched to return them in call order), the five outputs of
`GMMReg.forward(src, tgt)` in eval mode, and intermediates recomputed by the oracle AFTER it has
been checked bit-for-bit against the reference outputs (discrete indices; thin slices of the big
feature maps).  Weights are the closed-form fill of ogmm_amd/synth.py (not stored).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ogmm_oracle as O            # noqa: E402
from oracle.ref_harness import default_config, import_reference  # noqa: E402
from ogmm_amd import synth                      # noqa: E402

# name: (B, N, J, kind, first_pair, gnn_k, km_clusters)
CASES = {
    "partial_b2_n256_j16": (2, 256, 16, "partial", 100, 20, 128),
    "clean_b1_n1024_j16": (1, 1024, 16, "clean", 200, 20, 128),      # BASELINE configs[0]
    "partial_b2_n1024_j16": (2, 1024, 16, "partial", 0, 20, 128),    # first pairs of BASELINE configs[1]
    "partial_b1_n717_j128": (1, 717, 128, "partial", 300, 20, 128),  # the repo's own defaults (cfgs.py:21,34)
    "room_b1_n2048_j64": (1, 2048, 64, "room", 400, 20, 128),        # BASELINE configs[2]/[3] shape
    "partial_b3_n200_j8_k12": (3, 200, 8, "partial", 500, 12, 32),   # ragged: N not a multiple of anything
    # added with the round's later kernels: N between tile sizes (attention / overlap-block / EdgeConv partial tiles), k != 20 (EdgeConv's
    # run-time pooling), J = 32 on the grid-wide E/M (fused sweeps), 64 anchors; and room planes (exact kNN ties) at the headline shape
    "partial_b2_n1500_j32_k16": (2, 1500, 32, "partial", 600, 16, 64),
    "room_b2_n1024_j16": (2, 1024, 16, "room", 700, 20, 128),
    # round 3: the Sinkhorn early exit (lib/utils.py:99-102).  On unit-sphere clouds its batch-mean residual stays 50-800 x above the threshold;
    # clouds SCALED DOWN (8th entry) make the transport problem easy enough that the reference leaves its sweeps after 3-9 of 10 -- one case per
    # E/M kernel family of the HIP path (on-chip J = 16 / generic, resident, fused launch sequence is reached through OGMM_EM_RESIDENT=0, two-launch J > 64)
    "exit_partial_b2_n1024_j16": (2, 1024, 16, "partial", 800, 20, 128, 0.04),
    "exit_room_b1_n2048_j64": (1, 2048, 64, "room", 920, 20, 128, 0.03),
    "exit_partial_b2_n717_j128": (2, 717, 128, "partial", 1020, 20, 128, 0.03),
    "exit_partial_b3_n200_j8_k12": (3, 200, 8, "partial", 1110, 12, 32, 0.03),
    # round 4: a second, non-degenerate weight family (9th entry; synth.fill_state_dict(profile="sharp")): peaked attention (logits spanning +-10,
    # mean max probability 0.3-0.5 over the 128 anchors), overlap scores spanning (0.002, 0.999), wider BatchNorm statistics.  With the default
    # fill the attention is uniform to 1e-4 and every overlap score is 0.496 +- 0.003, a regime in which rounding Q / K / the scores cannot matter.
    "sharp_partial_b2_n1024_j16": (2, 1024, 16, "partial", 0, 20, 128, 1.0, "sharp"),      # BASELINE configs[1]
    "sharp_partial_b1_n2048_j64": (1, 2048, 64, "partial", 2000, 20, 128, 1.0, "sharp"),   # BASELINE configs[2] shape
    "sharp_partial_b1_n717_j128": (1, 717, 128, "partial", 300, 20, 128, 1.0, "sharp"),    # the repo's own defaults
    # round 5: room planes (exact kNN ties, the hardest discrete case) at BASELINE configs[3]'s per-cloud shape on the sharp family
    "sharp_room_b1_n2048_j64": (1, 2048, 64, "room", 3002, 20, 128, 1.0, "sharp"),
}


def run_reference(ref_mod, cfg, J, src, tgt, starts, profile="default"):
    net = ref_mod.GMMReg(512, J, cfg).eval()
    synth.fill_state_dict(net.state_dict(), profile=profile)
    calls = [0]
    real_randint = torch.randint

    def pinned(lo, hi, size, **kw):
        assert tuple(size) == (src.shape[0],) and hi == src.shape[2]
        out = starts[calls[0]].clone()
        calls[0] += 1
        return out

    torch.randint = pinned
    try:
        with torch.no_grad():
            out = net(src, tgt)
    finally:
        torch.randint = real_randint
    assert calls[0] == 6
    return out, {k: v.clone() for k, v in net.state_dict().items()}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_mod = import_reference()
    here = os.path.dirname(os.path.abspath(__file__))
    only = sys.argv[1:]                                  # names to (re)generate; default: all
    for name, case in CASES.items():
        if only and name not in only:
            continue
        (B, N, J, kind, first, k, M), scale = case[:7], (case[7] if len(case) > 7 else 1.0)
        profile = case[8] if len(case) > 8 else "default"
        cfg = default_config(n_clusters=J, gnn_k=k, km_clusters=M)
        src, tgt, R_gt, t_gt = synth.make_batch(first, B, N, kind)
        if scale != 1.0:
            src, tgt, t_gt = src * scale, tgt * scale, t_gt * scale
        starts = synth.fps_starts_for(first, B, N)
        (R, t, so, to, loss), P = run_reference(ref_mod, cfg, J, src, tgt, starts, profile)
        cap = {}
        with torch.no_grad():
            oR, ot, oso, oto, oloss = O.forward(P, cfg, src, tgt, starts, cap)
        for a, b in ((R, oR), (t, ot), (so, oso), (to, oto), (loss, oloss)):
            assert torch.equal(a, b), "oracle is not bit-identical to the reference on this machine"
        fx = dict(src=src.numpy(), tgt=tgt.numpy(), fps_starts=starts.numpy(), R_gt=R_gt.numpy(), t_gt=t_gt.numpy(),
                  R=R.numpy(), t=t.numpy(), src_o=so.numpy(), tgt_o=to.numpy(), loss=loss.numpy(),
                  meta=np.array([B, N, J, k, M, 512, 4]))
        if profile != "default":
            fx["profile"] = np.array(profile)
        for s in ("src", "tgt"):
            fx["knn_idx_" + s] = cap["knn_idx_" + s].numpy().astype(np.int16)
            for st in (0, 1, 2):
                fx["fps%d_%s" % (st, s)] = cap["fps%d_%s" % (st, s)].numpy().astype(np.int16)
            fx["fpsJ_" + s] = cap["fpsJ_" + s].numpy().astype(np.int16)
            fx["near_" + s] = cap["near_" + s].numpy().astype(np.int16)
            fx["pi_" + s] = cap["pi_" + s].numpy()
            fx["mu_" + s] = cap["mu_" + s].numpy()
            fx["gamma_rowsum_" + s] = cap["gamma_" + s].sum(-1).numpy()
            for key in ("emb", "pos", "ft", "f", "f2"):
                fx["%s8_%s" % (key, s)] = cap["%s_%s" % (key, s)][:, :8, :].numpy()
            fx["muf8_" + s] = cap["muf_" + s][:, :, :8].numpy()
            fx["wo_" + s] = cap["wo_" + s].numpy()
        fx["match_scores"] = cap["match_scores"].numpy()
        fx["sk_iters"] = np.array([cap["sk_iters_src"], cap["sk_iters_tgt"]], dtype=np.int32)      # [2, 10]: sweeps per E-step of the src / tgt call
        means = torch.cat([torch.stack(cap["sk_resid_" + s_]).mean(1) for s_ in ("src", "tgt")])
        fx["sk_margin"] = np.float32(((means - 1e-2).abs() / 1e-2).min().item())      # how close any decision came to the threshold (relative)
        print("   sweeps src %s tgt %s, closest decision %.3f of the threshold away" % (cap["sk_iters_src"], cap["sk_iters_tgt"], fx["sk_margin"]))
        assert fx["sk_margin"] > 0.03, "a decision within 3 % of the threshold: rounding differences between two correct implementations could flip it"
        if profile != "default":
            # what makes the family non-degenerate, recorded with the fixture: the attention's sharpness is a property of the weights + inputs that the
            # oracle reports (mean over queries of the largest probability, per transformer), the overlap scores' range is in the outputs themselves
            fx["attn_maxprob"] = np.array([cap["attn_maxprob_" + tr_] for tr_ in ("sattn1", "cattn", "sattn2")], dtype=np.float32)
            print("   attention mean max-probability %s, overlap scores %.4f ... %.4f" % (fx["attn_maxprob"], min(so.min(), to.min()), max(so.max(), to.max())))
            assert fx["attn_maxprob"].min() > 0.1 and min(so.min(), to.min()) < 0.02 and max(so.max(), to.max()) > 0.98
        if scale != 1.0 or profile != "default":
            # scaled-down clouds are often ill-conditioned: on many seeds the reference's own R moves by 3e-6 ... 1e-5 rad when only its thread count
            # changes (summation order).  A fixture has to be a case where "within 1e-5 of the reference" means something.
            torch.set_num_threads(1)
            with torch.no_grad():
                R1, _, so1, to1, _ = O.forward(P, cfg, src, tgt, starts)
            torch.set_num_threads(8)
            noise = O.rotation_error_rad(R1, R).max().item()
            fx["ref_thread_noise_o"] = np.float32(max((so1 - so).abs().max().item(), (to1 - to).abs().max().item()))
            print("   reference R, 1 thread against 8: %.2e rad; overlap scores %.2e" % (noise, fx["ref_thread_noise_o"]))
            fx["ref_thread_noise"] = np.float32(noise)
            # (the sharp family amplifies rounding: the reference's own R moves by 1e-6 ... 3e-6 rad between thread counts on most pairs, 9e-6 on some;
            #  a fixture may carry up to 3e-6 of it -- recorded with the fixture -- against the 1e-5 bar)
            assert noise < (3e-6 if profile != "default" else 1e-6), "ill-conditioned case: pick another seed / scale"
        path = os.path.join(here, name + ".npz")
        np.savez_compressed(path, **fx)
        print("%-28s %7.1f KB  R[0,0]=%+.6f loss=%.6f" % (name, os.path.getsize(path) / 1024, R[0, 0, 0], loss))


if __name__ == "__main__":
    main()
