"""GPU: the parity statement where it is thinnest -- the tail of the per-pair distribution, on three weight families.

(R, t) within 1e-5 of the reference is asserted pair by pair over hundreds of pairs; a pair beyond it must be one on which the REFERENCE's own fp32 result
is not defined to 1e-5 (tests/parity_util.py: its spread between host thread counts and against its fp64 evaluation), and this path must stay within a
small multiple of that spread.  Weight families: the closed-form default fill (near-uniform attention, overlap scores ~0.5: the degenerate regime all of
round 1-3's evidence was taken in), the closed-form "sharp" fill (attention logits spanning +-10, overlap scores spanning (0.002, 0.999)), and a
state_dict taken from 500 optimisation steps of this repo's own Trainer from PyTorch's default initialisation (the closest thing to the reference's
`optim_model.pt`, train.py:219-225, that exists without its dataset)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ogmm_oracle as O
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg, TERM_BUDGET
from parity_util import check_tail, distribution

pytestmark = pytest.mark.gpu


def _model(J, profile):
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    m = GMMReg(512, J, cfg)
    synth.fill_state_dict(m.state_dict(), profile=profile)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    return m.cuda().eval(), P, cfg


# (weight family, workload): (N, J, first pair, pairs, stated floor of pairs within 1e-5, cloud kind).  The floors are the measured counts of round 6
# (profiles/round6_parity*.txt) less three pairs at most.  N = 717 / J = 128 (the reference repo's own defaults) has 5.6 points per mixture component: the thinnest
# margin of all shapes.  The sharp windows are the FULL ones earlier profiles had run, misses included (configs[1] pairs 75 / 84 / 112; N = 717 pairs 309 / 357 /
# 422 / 348 ...), and the configs[3] room clouds (planes: exact kNN ties at rank k, the hardest discrete case) are run pair by pair on both weight families on the
# E/M launch sequence a 64-pair-per-GPU batch takes.
# Round 6: EVERY window is strict.  "cfg1b" (sharp pairs 128..319) was the window on which round 5 found the tail rule NOT to hold (pair 287 at 4.6 x the spread,
# pairs 160 / 301 beyond 1e-5 on well-conditioned pairs) and ran with strict=False.  tools/tail_bisect.py (the HIP path's stage results injected into the oracle one
# stage at a time) traced it to the fused InstanceNorm: its statistics were E[y^2] - mean^2 from fp32 partial sums, which loses |mean|^2 / var units in the last
# place, and the sharp family has channels at mean^2 / var ~ 10^3 ... 10^4.  With the sums in fp64 from the first value on (gemm_common.h) the window reads
# 187 of 192 within 1e-5, max 3.2e-5, and all five tail pairs within 1.5 x the reference's own spread; DESIGN.md section 2.
CASES = {
    ("default", "cfg1"): (1024, 16, 0, 256, 252, "partial"),          # measured 255; includes pair 128, round 3's worst (2.8e-5)
    ("default", "n717"): (717, 128, 300, 128, 125, "partial"),        # measured 128; includes pairs 334 and 413
    ("sharp", "cfg1"): (1024, 16, 0, 128, 120, "partial"),            # measured 123; 75, 84, 112 are in 64..127
    ("sharp", "cfg1b"): (1024, 16, 128, 192, 184, "partial"),         # measured 187
    ("sharp", "cfg1c"): (1024, 16, 320, 128, 121, "partial"),         # measured 124 (round 6's sweep beyond the asserted windows, profiles/round6_parity_sweep.txt: largest ratio 1.10)
    ("sharp", "n717"): (717, 128, 300, 128, 112, "partial"),          # measured 115: all 13 beyond are within 1.26 x the reference's own spread
    ("sharp", "cfg2"): (2048, 64, 2000, 16, 15, "partial"),
    ("default", "cfg3"): (2048, 64, 3000, 16, 15, "room"),
    ("sharp", "cfg3"): (2048, 64, 3000, 16, 15, "room"),
}


@pytest.mark.parametrize("profile,workload", list(CASES))
def test_every_pair_within_1e5_or_the_reference_itself_is_undefined_there(profile, workload, monkeypatch):
    N, J, first, B, floor, kind = CASES[(profile, workload)]
    if workload in ("cfg2", "cfg3"):
        monkeypatch.setenv("OGMM_EM_RESIDENT", "0")          # the launch sequence the full-size batch (256 pairs; 64 pairs per GPU) takes
    model, P, cfg = _model(J, profile)
    assert model.term_budget == TERM_BUDGET          # the shipped default, whatever it is
    label = "%s weights, %s (N=%d J=%d, %s pairs %d..%d)" % (profile, workload, N, J, kind, first, first + B - 1)
    r, t, o, inputs = distribution(model, P, cfg, first, B, N, kind, label=label)
    assert not model.fp16_overflowed()
    check_tail(label, r, t, inputs, P, cfg, first, floor)
    # the overlap scores are not part of the north star's bar; they are held to what the reference's own scores move by between thread counts
    # (default family 1e-5; sharp family: 2e-5 measured by tests/golden/make_golden.py, so 6e-5)
    assert o.max().item() < (1e-5 if profile == "default" else 6e-5)


def test_three_terms_everywhere_is_not_better_than_the_default_budget_at_n717():
    """the round-3 budget lost two pairs against three terms at N = 717 / J = 128 (126 against 128 of 128).  The default budget
    must not lose any: same pairs, both arithmetics, count within 1e-5."""
    N, J, first, B = 717, 128, 300, 128
    model, P, cfg = _model(J, "default")
    r_b, t_b, _, _ = distribution(model, P, cfg, first, B, N, "partial", label="n717 default budget")
    model.term_budget = {}
    r_3, t_3, _, _ = distribution(model, P, cfg, first, B, N, "partial", label="n717 three terms everywhere")
    within = lambda r, t: int(((r < 1e-5) & (t < 1e-5)).sum())  # noqa: E731
    assert within(r_b, t_b) >= within(r_3, t_3), (within(r_b, t_b), within(r_3, t_3))


def test_trained_weights_every_pair_against_the_oracle():
    """A state_dict after 500 steps of this repo's Trainer (PyTorch default init, the reference's crop sample chain on the device, 32 pairs of 717 points per
    step, Adam 1e-4: train.py's recipe) -- trained BatchNorm statistics, attention and overlap heads that have left their initial regime -- then every
    pair of the configs[1] batch against the oracle run live with the same weights."""
    from ogmm_amd import augment
    from ogmm_amd.trainer import Trainer
    dev = "cuda:0"
    torch.manual_seed(0)
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    model = GMMReg(512, 16, cfg).to(dev)
    tr = Trainer(model, lr=1e-4)
    pool = torch.stack([torch.from_numpy(synth._patch_cloud(np.random.Generator(np.random.PCG64(500 + i)), 1024)).float() for i in range(256)]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    first_loss = last = None
    for it in range(500):
        shapes = pool[torch.randint(0, pool.shape[0], (32,), generator=gen, device=dev)]
        smp = augment.crop_pipeline(shapes, augment.draw(32, 1024, 717, gen, dev), n_out=717)
        info = tr.step(smp["src_xyz"].transpose(1, 2).contiguous(), smp["tgt_xyz"].transpose(1, 2).contiguous(), smp["transform_gt"],
                       smp["src_overlap"], smp["tgt_overlap"])
        if it == 0:
            first_loss = float(info["loss"])
        last = info
    assert tr.skipped_steps <= 5 and float(last["loss"]) < 0.7 * first_loss, (tr.skipped_steps, first_loss, float(last["loss"]))
    model.eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # what regime the trained weights are in (diagnostic, from the oracle): attention sharpness, overlap-score range
    src, tgt, _, _ = synth.make_batch(0, 2, 1024, "partial")
    cap = {}
    with torch.no_grad():
        ref = O.forward(P, cfg, src, tgt, synth.fps_starts_for(0, 2, 1024), cap)
    print("TRAINED-WEIGHTS regime: loss %.3f -> %.3f, r_err %.1f deg; attention mean max-probability %s (uniform = 0.0078); overlap scores %.3f ... %.3f" % (
        first_loss, float(last["loss"]), float(last["r_err_deg"]), ["%.3f" % cap["attn_maxprob_" + t_] for t_ in ("sattn1", "cattn", "sattn2")],
        float(min(ref[2].min(), ref[3].min())), float(max(ref[2].max(), ref[3].max()))))
    label = "trained weights (500 steps), cfg1 (N=1024 J=16, pairs 0..63)"
    r, t, o, inputs = distribution(model, P, cfg, 0, 64, 1024, "partial", label=label)
    assert not model.fp16_overflowed()
    check_tail(label, r, t, inputs, P, cfg, 0, 60)
    # the same weights at the shape they were trained on (717 points: no multiple of any tile), 32 pairs
    label = "trained weights (500 steps), N=717 J=16, pairs 300..331"
    r, t, o, inputs = distribution(model, P, cfg, 300, 32, 717, "partial", label=label)
    assert not model.fp16_overflowed()
    check_tail(label, r, t, inputs, P, cfg, 300, 29)
