"""GPU: GMMReg.forward on the HIP path against (a) the golden fixtures produced by the reference itself and
(b) the CPU oracle run live on the same seeded inputs.  Target (BASELINE.json north_star): R within 1e-5 rad,
t within 1e-5 units; discrete intermediates (kNN graph, FPS chains) identical."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import golden_names
from oracle import ogmm_oracle as O
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg

pytestmark = pytest.mark.gpu

R_TOL = 1e-5      # rad
T_TOL = 1e-5      # cloud units
O_TOL = 5e-6      # overlap scores in (0,1)
LOSS_TOL = 5e-5


def build(cfg, J, precision="f16x3", profile="default"):
    m = GMMReg(512, J, cfg)
    m.precision = precision
    synth.fill_state_dict(m.state_dict(), profile=profile)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    return m.cuda().eval(), P


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("name", golden_names())
def test_forward_matches_reference_golden(golden, name, precision):
    fx = golden(name)
    B, N, J, k, M, D, H = [int(v) for v in fx["meta"]]
    cfg = Namespace(gnn_k=k, num_heads=H, km_clusters=M, overlap_radius=0.035, n_clusters=J)
    model, _ = build(cfg, J, precision, str(fx["profile"]) if "profile" in fx else "default")      # sharp_*: the second weight family (peaked attention, saturated scores)
    src, tgt = torch.from_numpy(fx["src"]).cuda(), torch.from_numpy(fx["tgt"]).cuda()
    with torch.no_grad():
        R, t, so, to, loss = model(src, tgt, fps_starts=torch.from_numpy(fx["fps_starts"]), capture=True)
    cap = model.last_intermediates
    # kNN graph: identical sorted distance rows and identical neighbour sets (order inside exact ties may differ)
    idx = cap["knn_idx"].cpu().long()
    ref_idx = torch.from_numpy(np.concatenate([fx["knn_idx_src"], fx["knn_idx_tgt"]], 0).astype(np.int64))
    xyz = torch.cat([torch.from_numpy(fx["src"]), torch.from_numpy(fx["tgt"])], 0).transpose(1, 2).contiguous()
    d = O.sq_dist_expanded(xyz, xyz)
    assert torch.equal(torch.gather(d, 2, idx), torch.gather(d, 2, ref_idx))
    set_differs = (idx.sort(-1)[0] != ref_idx.sort(-1)[0]).any(-1)
    assert not bool(set_differs.any()), "neighbour sets differ from the reference in %d rows" % int(set_differs.sum())
    ids = cap["fps_anchor"].cpu().numpy()
    for st in range(3):
        assert np.array_equal(ids[st, :B], fx["fps%d_src" % st]) and np.array_equal(ids[st, B:], fx["fps%d_tgt" % st])
    idj = cap["fps_J"].cpu().numpy()
    assert np.array_equal(idj[:B], fx["fpsJ_src"]) and np.array_equal(idj[B:], fx["fpsJ_tgt"])
    dk = d.topk(k + 1, dim=-1, largest=False)[0]
    rep = {"rows_with_rank_k_tie": int((dk[:, :, k - 1] == dk[:, :, k]).sum())}
    for key, g in (("emb", cap["emb"]), ("ft", cap["ft"]), ("f", cap["f"]), ("f2", cap["f2"])):
        got = g.view(2 * B, N, D)[:, :, :8].transpose(1, 2).cpu().numpy()
        ref = np.concatenate([fx[key + "8_src"], fx[key + "8_tgt"]], 0)
        rep[key] = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
    # Sinkhorn early exit (lib/utils.py:99-102): every E-step of the src call and of the tgt call ran exactly the sweeps the reference ran -- all 10
    # on the unit-sphere fixtures, 3-9 on the scaled-down `exit_*` ones; the residual is measured on every kernel family (no NaN in sweeps that ran)
    want = fx["sk_iters"] if "sk_iters" in fx else np.full((2, 10), 10)
    got_sweeps = cap["sinkhorn_sweeps"].cpu().numpy()
    assert np.array_equal(got_sweeps, want), (got_sweeps, want)
    resid = cap["sinkhorn_resid"].cpu().numpy().reshape(2, B, 10, 10)
    for g in range(2):
        for it in range(10):
            assert not np.isnan(resid[g, :, it, :want[g, it]]).any() and np.isnan(resid[g, :, it, want[g, it]:]).all()
    margin = model.sinkhorn_exit_margin()
    assert (margin <= 1.0) == bool(want.min() < 10)
    rep["sinkhorn_margin"] = margin
    rep["R"] = O.rotation_error_rad(R.cpu(), torch.from_numpy(fx["R"])).max().item()
    rep["t"] = O.translation_error(t.cpu(), torch.from_numpy(fx["t"])).max().item()
    rep["o"] = float(max(np.abs(so.cpu().numpy() - fx["src_o"]).max(), np.abs(to.cpu().numpy() - fx["tgt_o"]).max()))
    rep["loss"] = abs(loss.item() - float(fx["loss"]))
    print("PARITY", precision, name, " ".join("%s=%.2e" % kv for kv in rep.items()))
    assert not model.fp16_overflowed()
    for key in ("emb", "ft", "f", "f2"):
        assert rep[key] < 2e-5, (key, rep)
    assert rep["R"] < R_TOL and rep["t"] < T_TOL, rep
    # sharp family: the head's gain makes the reference's OWN scores move by 2e-5 between 1 and 8 host threads (recorded with the fixture by
    # make_golden.py); the bar on the scores is then 3 x that, the bar on (R, t) stays the north star's 1e-5
    o_tol = max(O_TOL, min(3.0 * float(fx["ref_thread_noise_o"]), 7.5e-5)) if "profile" in fx else O_TOL          # (capped: a regenerated fixture cannot loosen it further)
    assert rep["o"] < o_tol and rep["loss"] < LOSS_TOL, rep


def test_forward_matches_oracle_live_batch():
    """BASELINE configs[1] shape at a batch the CPU oracle finishes in seconds; also checks batch invariance
    (pairs are independent: SURVEY.md 8e) by comparing against a B=1 run of one of the pairs."""
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    B, N = 6, 1024
    model, P = build(cfg, 16)
    src, tgt, _, _ = synth.make_batch(1000, B, N, "partial")
    starts = synth.fps_starts_for(1000, B, N)
    with torch.no_grad():
        R, t, so, to, loss = model(src.cuda(), tgt.cuda(), fps_starts=starts)
        Ro, to_, soo, too, losso = O.forward(P, cfg, src, tgt, starts)
        R1, t1, so1, _, _ = model(src[3:4].cuda(), tgt[3:4].cuda(), fps_starts=starts[:, 3:4])
    assert O.rotation_error_rad(R.cpu(), Ro).max().item() < R_TOL
    assert O.translation_error(t.cpu(), to_).max().item() < T_TOL
    assert (so.cpu() - soo).abs().max().item() < O_TOL and (to.cpu() - too).abs().max().item() < O_TOL
    assert abs(loss.item() - losso.item()) < LOSS_TOL
    assert O.rotation_error_rad(R1.cpu(), R[3:4].cpu()).max().item() < 2e-6
    assert (so1.cpu() - so[3:4].cpu()).abs().max().item() < 2e-6


@pytest.mark.parametrize("D,heads", [(256, 4), (1024, 4), (512, 8), (256, 2)])
def test_other_embedding_sizes_and_head_counts_match_the_oracle(D, heads):
    """emb_dims / num_heads away from the reference's defaults (512 / 4): the fused attention is built for 128-wide heads and several fusions for 512 channels;
    every other configuration takes the general kernels and must land on the same results."""
    cfg = Namespace(gnn_k=20, num_heads=heads, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    m = GMMReg(D, 16, cfg)
    synth.fill_state_dict(m.state_dict())
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    src, tgt, _, _ = synth.make_batch(0, 3, 512, "partial")
    starts = synth.fps_starts_for(0, 3, 512)
    with torch.no_grad():
        R, t, so, to, loss = m(src.cuda(), tgt.cuda(), fps_starts=starts)
        Ro, to_, soo, too, losso = O.forward(P, cfg, src, tgt, starts)
    assert O.rotation_error_rad(R.cpu(), Ro).max().item() < R_TOL and O.translation_error(t.cpu(), to_).max().item() < T_TOL
    assert (so.cpu() - soo).abs().max().item() < O_TOL and (to.cpu() - too).abs().max().item() < O_TOL and abs(loss.item() - losso.item()) < LOSS_TOL
    assert not m.fp16_overflowed()


def test_forward_draws_fps_starts_like_the_reference():
    """With fps_starts=None the six draws come from torch's global CPU generator in the reference's order."""
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=64, overlap_radius=0.035, n_clusters=16)
    model, _ = build(cfg, 16)
    src, tgt, _, _ = synth.make_batch(7, 2, 256, "clean")
    torch.manual_seed(99)
    expect = O.draw_fps_starts(2, 256)
    torch.manual_seed(99)
    with torch.no_grad():
        a = model(src.cuda(), tgt.cuda(), capture=True)
        ids_a = model.last_intermediates["fps_anchor"].clone()
        b = model(src.cuda(), tgt.cuda(), fps_starts=expect, capture=True)
    assert torch.equal(ids_a, model.last_intermediates["fps_anchor"])
    assert torch.equal(a[0], b[0])


def test_full_size_properties_config1():
    """BASELINE configs[1] at full size (B=64): no oracle at this size in the time budget, so size-independent
    properties: rotations are proper, overlap in (0,1), finite loss, and sharding the batch does not change results."""
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    model, _ = build(cfg, 16)
    B, N = 64, 1024
    src, tgt, _, _ = synth.make_batch(0, B, N, "partial")
    starts = synth.fps_starts_for(0, B, N)
    with torch.no_grad():
        R, t, so, to, loss = model(src.cuda(), tgt.cuda(), fps_starts=starts)
        Rh, th, soh, _, _ = model(src[32:].cuda(), tgt[32:].cuda(), fps_starts=starts[:, 32:])
    Rd = R.double().cpu()
    assert (Rd @ Rd.transpose(1, 2) - torch.eye(3, dtype=torch.double)).abs().max().item() < 1e-5
    assert (torch.det(Rd) - 1).abs().max().item() < 1e-5
    assert 0 < float(so.min()) and float(so.max()) < 1 and torch.isfinite(loss)
    assert O.rotation_error_rad(Rh.cpu(), R[32:].cpu()).max().item() < 2e-6
    assert (soh.cpu() - so[32:].cpu()).abs().max().item() < 2e-6


def _distribution(name, model, P, cfg, first, B, N, kind, threads=16):
    """every pair of a batch against the oracle (run in chunks of 8 pairs on the host): per-pair R / t / overlap errors"""
    src, tgt, _, _ = synth.make_batch(first, B, N, kind)
    starts = synth.fps_starts_for(first, B, N)
    with torch.no_grad():
        got = [x.cpu() for x in model(src.cuda(), tgt.cuda(), fps_starts=starts)[:4]]
    old = torch.get_num_threads()
    torch.set_num_threads(min(threads, os.cpu_count() or threads))
    r, t, o = [], [], []
    try:
        for a in range(0, B, 8):
            e = min(B, a + 8)
            with torch.no_grad():
                ref = O.forward(P, cfg, src[a:e], tgt[a:e], starts[:, a:e])
            r.append(O.rotation_error_rad(got[0][a:e], ref[0]))
            t.append(O.translation_error(got[1][a:e], ref[1]))
            o.append(torch.maximum((got[2][a:e] - ref[2]).abs().amax(1), (got[3][a:e] - ref[3]).abs().amax(1)))
    finally:
        torch.set_num_threads(old)
    r, t, o = torch.cat(r), torch.cat(t), torch.cat(o)
    edges = [0.0, 3e-7, 1e-6, 3e-6, 1e-5, 1.0]
    hist = " ".join("<%.0e:%d" % (hi, int(((r >= lo) & (r < hi)).sum())) for lo, hi in zip(edges[:-1], edges[1:]))
    print("PARITY-DISTRIBUTION %s: %d pairs  R max %.2e median %.2e [%s]  t max %.2e  overlap max %.2e" % (name, B, r.max(), r.median(), hist, t.max(), o.max()))
    return r, t, o


@pytest.mark.gpu
def test_full_batch_matches_oracle_on_every_pair():
    """BASELINE configs[1] at its full size (64 pairs, N = 1024, J = 16) -- the batch bench.py times, on the engines it times (the LDS-DMA engines with
    their fused forms and the per-layer term budget only run at this size): EVERY pair of the 64-pair forward against the CPU oracle, not a sample.
    The E/M + matching head is ill-conditioned, so the tail decides: max over the pairs of R and t within 1e-5 (north_star)."""
    B, N, J = 64, 1024, 16
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model, P = build(cfg, J)
    r, t, o = _distribution("configs[1] N=1024 J=16", model, P, cfg, 0, B, N, "partial")
    assert r.max().item() < 1e-5 and t.max().item() < 1e-5 and o.max().item() < 1e-5
    assert not model.fp16_overflowed()


@pytest.mark.gpu
def test_config2_shape_on_the_large_batch_path_matches_oracle_on_every_pair(monkeypatch):
    """BASELINE configs[2] (unseen-category-like partial clouds, N = 2048, J = 64, batch 256): 16 pairs through the code path a 256-pair batch takes --
    the grid-wide E/M launch sequence (a 512-cloud grid does not fit the resident kernel: OGMM_EM_RESIDENT=0 forces the same choice here), the
    large-shape GEMM engines (65536 rows = 256 row tiles), the matrix-core cluster feature means -- every pair against the oracle."""
    monkeypatch.setenv("OGMM_EM_RESIDENT", "0")
    B, N, J = 16, 2048, 64
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model, P = build(cfg, J)
    r, t, o = _distribution("configs[2] N=2048 J=64 (large-batch path)", model, P, cfg, 2000, B, N, "partial")
    assert r.max().item() < 1e-5 and t.max().item() < 1e-5 and o.max().item() < 1e-5


@pytest.mark.gpu
def test_repo_default_shape_matches_oracle_on_every_pair():
    """The reference repo's own defaults (N = 717, J = 128: J close to N makes the E/M ill-conditioned -- the shape with the thinnest margin): 32 pairs."""
    B, N, J = 32, 717, 128
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model, P = build(cfg, J)
    r, t, o = _distribution("repo defaults N=717 J=128", model, P, cfg, 300, B, N, "partial")
    assert r.max().item() < 1e-5 and t.max().item() < 1e-5 and o.max().item() < 1e-5


@pytest.mark.gpu
def test_full_size_properties_config2():
    """BASELINE configs[2] at its full size (256 pairs of 2048 points, J = 64): no oracle at this size in the time budget, so size-independent
    properties: proper rotations, overlap scores in (0,1), finite loss, every E-step ran the reference's 10 sweeps, and a 64-pair shard of the batch
    gives what the full batch gives for those pairs (pairs are independent; the Sinkhorn exit -- the only coupling -- does not fire here)."""
    B, N, J = 256, 2048, 64
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model, _ = build(cfg, J)
    src, tgt, _, _ = synth.make_batch(2000, B, N, "partial")
    starts = synth.fps_starts_for(2000, B, N)
    with torch.no_grad():
        R, t, so, to, loss = model(src.cuda(), tgt.cuda(), fps_starts=starts, capture=True)
        assert bool((model.last_intermediates["sinkhorn_sweeps"] == 10).all())
        Rh, th, soh, _, _ = model(src[64:128].cuda(), tgt[64:128].cuda(), fps_starts=starts[:, 64:128])
    Rd = R.double().cpu()
    assert (Rd @ Rd.transpose(1, 2) - torch.eye(3, dtype=torch.double)).abs().max().item() < 1e-5
    assert (torch.det(Rd) - 1).abs().max().item() < 1e-5
    assert 0 < float(so.min()) and float(so.max()) < 1 and 0 < float(to.min()) and float(to.max()) < 1 and torch.isfinite(loss)
    assert O.rotation_error_rad(Rh.cpu(), R[64:128].cpu()).max().item() < 4e-6
    assert (soh.cpu() - so[64:128].cpu()).abs().max().item() < 4e-6
    assert not model.fp16_overflowed()


# layers that the reduced mode (precision = "f16") multiplies with ONE binary16 term per operand on the large-shape engine (ops.gemm_nt single_term);
# the EdgeConv kernel, the attention kernel and the small anchor-side projections keep their split terms
def _f16_mode_policy(name):
    if name.startswith("emd.conv") and name != "emd.conv5":
        return "x3"
    if name.endswith((".attn.qk", ".attn.pv", ".attn.proj.1", ".attn.proj.2")):
        return "x3"
    return "x1"


@pytest.mark.gpu
def test_reduced_precision_mode_against_the_emulating_oracle():
    """BASELINE configs[2] is quoted in bf16.  The labelled reduced mode here is precision = "f16": one binary16 term per operand (11 significand bits
    >= bf16's 8) in the large GEMMs, fp32 accumulation.  It cannot meet 1e-5, so its tolerance is DERIVED: the CPU oracle with the
    same operand rounding (oracle/split_emulation.py, mode "x1" on the same layers) gives the deviation that this arithmetic causes in the
    reference's own algorithm; the HIP path in that mode must stay within 4x of it (max and median over 16 pairs of the configs[2] shape) --
    rounding noise is chaotic pair by pair, so the comparison is between distributions, both measured against the exact oracle."""
    from oracle import split_emulation as E
    B, N, J = 16, 2048, 64
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model, P = build(cfg, J, precision="f16")
    src, tgt, _, _ = synth.make_batch(2000, B, N, "partial")
    starts = synth.fps_starts_for(2000, B, N)
    with torch.no_grad():
        got = [x.cpu() for x in model(src.cuda(), tgt.cuda(), fps_starts=starts)[:2]]
    old = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    r_hip, r_emu, t_hip, t_emu = [], [], [], []
    try:
        for a in range(0, B, 8):
            with torch.no_grad():
                exact = O.forward(P, cfg, src[a:a + 8], tgt[a:a + 8], starts[:, a:a + 8])
                with E.policy(_f16_mode_policy):
                    emu = O.forward(P, cfg, src[a:a + 8], tgt[a:a + 8], starts[:, a:a + 8])
            r_hip.append(O.rotation_error_rad(got[0][a:a + 8], exact[0])); t_hip.append(O.translation_error(got[1][a:a + 8], exact[1]))
            r_emu.append(O.rotation_error_rad(emu[0], exact[0])); t_emu.append(O.translation_error(emu[1], exact[1]))
    finally:
        torch.set_num_threads(old)
    r_hip, r_emu, t_hip, t_emu = torch.cat(r_hip), torch.cat(r_emu), torch.cat(t_hip), torch.cat(t_emu)
    print("PARITY f16 (reduced) on configs[2] shape, 16 pairs, against the exact oracle: HIP R max %.2e median %.2e t max %.2e | emulating oracle R max %.2e median %.2e t max %.2e"
          % (r_hip.max(), r_hip.median(), t_hip.max(), r_emu.max(), r_emu.median(), t_emu.max()))
    assert r_emu.max().item() > 1e-5, "the emulation shows no effect: nothing tested"
    assert r_hip.max().item() < 4 * r_emu.max().item() and r_hip.median().item() < 4 * r_emu.median().item()
    assert t_hip.max().item() < 4 * t_emu.max().item()
    assert r_hip.median().item() > 0.1 * r_emu.median().item(), "the reduced mode is far more accurate than its emulation: it does not run the labelled arithmetic"


@pytest.mark.gpu
@pytest.mark.parametrize("profile,sample", [("default", (0, 9, 18, 27, 36, 45, 54, 63)), ("sharp", (5, 23, 41, 59))])
def test_configs3_at_its_full_per_gpu_size(profile, sample):
    """BASELINE configs[3] at the size ONE GPU takes of its 512-pair batch: 64 room-cloud pairs x 2048 points, J = 64, in ONE forward (the
    suite had 16 pairs on the forced launch sequence and a 3-pair sample in the bench).  Every pair: R orthonormal with det +1, finite t, scores in (0, 1), all E-steps
    ran their ten sweeps (so the per-call batch-mean exit did not couple the clouds and a pair is the oracle's single-pair computation); a strided sample of the
    pairs against the oracle by the tail rule of tests/parity_util.py."""
    from parity_util import check_tail
    B, N, J, first = 64, 2048, 64, 3000
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model, P = build(cfg, J, profile=profile)
    src, tgt, _, _ = synth.make_batch(first, B, N, "room")
    starts = synth.fps_starts_for(first, B, N)
    with torch.no_grad():
        R, t, so, to_, loss = model(src.cuda(), tgt.cuda(), fps_starts=starts, capture=True)
    assert not model.fp16_overflowed()
    sweeps = model.last_intermediates["sinkhorn_sweeps"].cpu()
    assert int(sweeps.min()) == 10 and int(sweeps.max()) == 10, sweeps
    R, t, so, to_ = R.cpu(), t.cpu(), so.cpu(), to_.cpu()
    eye = torch.eye(3)[None].expand(B, 3, 3)
    assert torch.isfinite(R).all() and torch.isfinite(t).all() and torch.isfinite(loss).all()
    assert (R @ R.transpose(1, 2) - eye).abs().max() < 1e-5 and (torch.linalg.det(R) - 1).abs().max() < 1e-5
    assert float(so.min()) > 0 and float(so.max()) < 1 and float(to_.min()) > 0 and float(to_.max()) < 1
    ids = list(sample)
    old = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    try:
        r, tt, oo = [], [], []
        for i in ids:
            with torch.no_grad():
                ref = O.forward(P, cfg, src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1])
            r.append(O.rotation_error_rad(R[i:i + 1], ref[0])); tt.append(O.translation_error(t[i:i + 1], ref[1]))
            oo.append(max((so[i:i + 1] - ref[2]).abs().max().item(), (to_[i:i + 1] - ref[3]).abs().max().item()))
    finally:
        torch.set_num_threads(old)
    r, tt = torch.cat(r), torch.cat(tt)
    label = "%s weights, configs[3] per-GPU batch (64 room pairs x 2048, J=64), sampled pairs %s" % (profile, [first + i for i in ids])
    print("PARITY-DISTRIBUTION %s: R max %.2e median %.2e  t max %.2e  overlap max %.2e" % (label, r.max(), r.median(), tt.max(), max(oo)))
    sel = torch.tensor(ids)
    # (check_tail numbers its pairs first + position: hand it the sampled pairs as a contiguous block of their own)
    check_tail(label, r, tt, (src[sel], tgt[sel], starts[:, sel]), P, cfg, 0, len(ids) - 1)
    assert max(oo) < (1e-5 if profile == "default" else 6e-5)


@pytest.mark.gpu
def test_largest_supported_cloud_and_input_validation():
    """N = 4096 (the FPS / kNN kernels keep a whole cloud on chip: their documented ceiling), J = 64, against the CPU oracle; and the
    argument checks of the forward: wrong dtype, N_src != N_tgt, more neighbours / anchors / clusters than points."""
    from ogmm_amd._lib import OgmmError
    B, N, J = 1, 4096, 64
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model = GMMReg(512, J, cfg)
    synth.fill_state_dict(model.state_dict())
    P = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to("cuda:0").eval()
    src, tgt, _, _ = synth.make_batch(900, B, N, "room")
    starts = synth.fps_starts_for(900, B, N)
    with torch.no_grad():
        R, t, so, to_, loss = model(src.cuda(), tgt.cuda(), fps_starts=starts)
        Ro, to, soo, too, losso = O.forward(P, cfg, src, tgt, starts)
    r, tt = O.rotation_error_rad(R.cpu(), Ro).max().item(), O.translation_error(t.cpu(), to).max().item()
    print("PARITY f16x3 live N=4096 J=64: R=%.2e t=%.2e o=%.2e" % (r, tt, (so.cpu() - soo).abs().max().item()))
    assert r < R_TOL and tt < T_TOL and (so.cpu() - soo).abs().max().item() < 1e-5
    x = torch.zeros(2, 3, 64, device="cuda:0")
    for bad in ((x.double(), x.double()), (x, torch.zeros(2, 3, 65, device="cuda:0")), (x[:, :, :10], x[:, :, :10]),
                (torch.zeros(2, 4, 64, device="cuda:0"),) * 2):
        with pytest.raises(OgmmError):
            model(*bad)
    # the kernels' ceilings are argument errors of the forward, not launch errors from inside it
    y = torch.zeros(1, 3, 512, device="cuda:0")
    for k_, J_ in ((1, 16), (40, 16), (20, 200)):
        cfg_ = Namespace(gnn_k=k_, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J_)
        with pytest.raises(OgmmError, match="gnn_k"):
            GMMReg(512, J_, cfg_).to("cuda:0").eval()(y, y)


@pytest.mark.gpu
def test_hip_graph_replay_is_bit_identical():
    """GMMReg.capture_graph: the ~140 launches of the eval forward (two streams) recorded once and replayed with one hipGraphLaunch."""
    B, N, J = 3, 512, 16
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
    model = GMMReg(512, J, cfg)
    synth.fill_state_dict(model.state_dict())
    model = model.to("cuda:0").eval()
    run = model.capture_graph(B, N)
    for first in (10, 50):
        src, tgt, _, _ = synth.make_batch(first, B, N, "partial")
        starts = synth.fps_starts_for(first, B, N)
        with torch.no_grad():
            eager = [t.clone() for t in model(src.cuda(), tgt.cuda(), fps_starts=starts)]
            replay = run(src.cuda(), tgt.cuda(), starts.cuda())
        for a, b in zip(eager, replay):
            assert torch.equal(a, b)
    with pytest.raises(Exception):
        run(torch.zeros(2, 3, N, device="cuda:0"), torch.zeros(2, 3, N, device="cuda:0"))
    # a SECOND capture of the same shape must not share (or find "clean") the first graph's statistics / side-input buffers: each closure
    # owns a set that was zeroed eagerly before its capture; both graphs replay correctly in any order, also when the second one runs first
    run2 = model.capture_graph(B, N)
    assert run2.workspace is not run.workspace and run2.workspace["stats3"].data_ptr() != run.workspace["stats3"].data_ptr()
    src, tgt, _, _ = synth.make_batch(90, B, N, "partial")
    starts = synth.fps_starts_for(90, B, N)
    with torch.no_grad():
        eager = [t.clone() for t in model(src.cuda(), tgt.cuda(), fps_starts=starts)]
        second = [t.clone() for t in run2(src.cuda(), tgt.cuda(), starts.cuda())]
        first = [t.clone() for t in run(src.cuda(), tgt.cuda(), starts.cuda())]
    for a, b, c in zip(eager, second, first):
        assert torch.equal(a, b) and torch.equal(a, c)
    # (no fill of the 16.8 MB-class side-input buffer is recorded: the graph's buffers were zeroed before the capture)
    assert float(run2.workspace["stats3"].abs().sum()) == 0.0          # self-cleaned behind the replay


@pytest.mark.gpu
def test_eval_forward_does_not_pass_an_fp16_range_overflow_silently():
    """An activation beyond +-65504 is clamped by the fp16x3 engines; results built on it are not the reference's.  Policy "sync": the forward raises
    itself; "deferred" (default): no host synchronisation, the flag arrives behind the forward and the NEXT call (or fp16_overflowed()) reports it;
    the exact-fp32 engine has no such range and runs the same weights without complaint."""
    from ogmm_amd._lib import OgmmError
    B, N = 2, 512
    src, tgt, _, _ = synth.make_batch(5, B, N, "partial")
    starts = synth.fps_starts_for(5, B, N)
    src, tgt = src.cuda(), tgt.cuda()

    def model_with_large_activations(precision):
        m = GMMReg(512, 16, Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, precision=precision))
        synth.fill_state_dict(m.state_dict())
        with torch.no_grad():
            m.state_dict()["conv1.net.0.weight"].mul_(1.0e5)          # conv1's hidden map ~1e5: beyond binary16
        return m.cuda().eval()
    m = model_with_large_activations("f16x3")
    m.overflow_policy = "sync"
    with torch.no_grad(), pytest.raises(OgmmError, match="65504"):
        m(src, tgt, fps_starts=starts)
    m = model_with_large_activations("f16x3")
    assert m.overflow_policy == "deferred"
    with torch.no_grad():
        m(src, tgt, fps_starts=starts)          # returns (nothing waited for) ...
        torch.cuda.synchronize()
        with pytest.raises(OgmmError, match="earlier forward"):
            m(src, tgt, fps_starts=starts)      # ... and the next call reports it
    m = model_with_large_activations("f16x3")
    with torch.no_grad():
        m(src, tgt, fps_starts=starts)
    assert m.fp16_overflowed() and not m.fp16_overflowed()
    m = model_with_large_activations("f32")
    with torch.no_grad():
        out = m(src, tgt, fps_starts=starts)
    assert torch.isfinite(out[0]).all() and not m.fp16_overflowed()



def test_consecutive_forwards_are_bit_reproducible_without_a_synchronisation():
    """Round 5: the forward no longer synchronises the host (its anchor draws travel through pinned memory), so consecutive forwards are in flight together.
    Six forwards on resident batches of alternating shapes (the persistent workspace is keyed per shape), enqueued back to back, must give exactly the outputs
    of the same forwards run one at a time -- FPS chains, statistics buffers and side streams included."""
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    model, _ = build(cfg, 16)
    batches = []
    for i, (B, N) in enumerate(((6, 1024), (4, 717), (6, 1024), (4, 717), (32, 1024), (32, 1024))):
        src, tgt, _, _ = synth.make_batch(40 + 10 * i, B, N, "partial")
        batches.append((src.cuda(), tgt.cuda(), synth.fps_starts_for(40 + 10 * i, B, N)))
    torch.cuda.synchronize()
    outs = {}
    for sync in (True, False, False):
        res = []
        with torch.no_grad():
            for s, t, st in batches:
                res.append([x.clone() for x in model(s, t, fps_starts=st)])
                if sync:
                    torch.cuda.synchronize()
        torch.cuda.synchronize()
        if sync:
            outs = res
        else:
            for a, b in zip(outs, res):
                for x, y in zip(a, b):
                    assert torch.equal(x, y)
    assert not model.fp16_overflowed()


def test_pipelined_head_gives_the_same_outputs_over_consecutive_forwards():
    """GMMReg.pipeline_head (round 5): the head of a forward -- kNN graph, positional front end, all FPS chains -- on its own streams, not waiting for the previous
    forward's tail.  Consecutive forwards on different resident batches, enqueued without a synchronisation in between, must give exactly the outputs of the
    serial order (same kernels, same inputs: bit-identical), also when the batches alternate between shapes (the workspace is keyed per shape), three times over:
    the FPS chains now run beside the previous forward's GEMMs, the schedule in which they were NOT reproducible before the library lost its packed-fp32
    instructions (HISTORY.md section 4)."""
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=16)
    model, _ = build(cfg, 16)
    batches = []
    for i, (B, N) in enumerate(((6, 1024), (4, 717), (6, 1024), (4, 717), (32, 1024), (32, 1024))):
        src, tgt, _, _ = synth.make_batch(40 + 10 * i, B, N, "partial")
        batches.append((src.cuda(), tgt.cuda(), synth.fps_starts_for(40 + 10 * i, B, N)))
    torch.cuda.synchronize()
    ref = None
    for flag in (False, True, True, True):
        model.pipeline_head = flag
        with torch.no_grad():
            res = [[x.clone() for x in model(s, t, fps_starts=st)] for s, t, st in batches]          # no synchronisation between the forwards
        torch.cuda.synchronize()
        if ref is None:
            ref = res
            continue
        for a, b in zip(ref, res):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
    assert not model.fp16_overflowed()
