"""Stage-by-stage bisection of the parity tail (GPU box).  VERDICT round 5, next-1 / ADVICE round 5, medium.

    python tools/tail_bisect.py [--profile sharp] [--workload cfg1] [--first 256] [--pairs 64] [--precision f16x3] [--ids 287,266] [--all]

One batch through the HIP forward with capture=True.  For every listed pair (default: every pair of the batch beyond 1e-5) the HIP path's captured
stage results are INJECTED into the oracle one stage at a time (oracle.forward(inject=...)): everything downstream of the injected stage is then
evaluated in the reference's arithmetic, so

    d(inject X) = | R(oracle downstream of HIP's X) - R(oracle) |

is what the HIP path's difference UP TO AND INCLUDING stage X does to the result.  Reading the row from left to right (emb, ft, f, f-for-the-overlap-chain
only, o, f2, E/M, muf, and the HIP result itself) the column at which the number first reaches the HIP path's own distance is the stage that causes it.
Second table: how far each captured stage is from the oracle's fp64 evaluation, beside how far the reference's own fp32 is from it ("is this path LESS
accurate than the reference at that stage, or only differently rounded?"), and three single-kernel checks on the ORACLE's inputs (the E/M kernel on the
oracle's overlap scores, the feature means on the oracle's gamma / f2, the matching + rigid solve on the oracle's mu / muf)."""
import argparse
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import ogmm_oracle as O  # noqa: E402
from ogmm_amd import ops, synth  # noqa: E402
from ogmm_amd.gmmreg import GMMReg  # noqa: E402

WORK = {"cfg1": (1024, 16, "partial"), "cfg2": (2048, 64, "partial"), "cfg3": (2048, 64, "room"), "n717": (717, 128, "partial")}


def d_rot(a, b):
    return O.rotation_error_rad(a.double(), b.double()).max().item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile", default="sharp")
    ap.add_argument("--workload", default="cfg1")
    ap.add_argument("--first", type=int, default=256)
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--ids", default="")
    ap.add_argument("--all", action="store_true", help="every pair of the batch, not only the tail (gives the typical row to compare a tail row with)")
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    N, J, kind = WORK[args.workload]
    cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035, n_clusters=J)
    model = GMMReg(512, J, cfg)
    synth.fill_state_dict(model.state_dict(), profile=args.profile)
    P = {k: v.clone() for k, v in model.state_dict().items()}
    P64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P.items()}
    model = model.cuda().eval()
    model.precision = args.precision
    first, B = args.first, args.pairs
    D = 512
    src, tgt, _, _ = synth.make_batch(first, B, N, kind)
    starts = synth.fps_starts_for(first, B, N)
    torch.set_num_threads(args.threads)
    with torch.no_grad():
        out = model(src.cuda(), tgt.cuda(), fps_starts=starts, capture=True)
        torch.cuda.synchronize()
    g = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in model.last_intermediates.items()}
    hipR, hipT = out[0].cpu(), out[1].cpu()
    assert not model.fp16_overflowed()
    C = 2 * B

    def feat(key, i):          # HIP [C*N, D] point-major -> the oracle's ([1,D,N] src, [1,D,N] tgt) of pair i
        x = g[key].view(C, N, -1)
        return x[i].t().contiguous()[None], x[B + i].t().contiguous()[None]

    refs, caps = [], []
    for i in range(B):
        with torch.no_grad():
            cap = {}
            refs.append(O.forward(P, cfg, src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1], cap=cap))
            caps.append(cap)
    r = torch.tensor([d_rot(hipR[i:i + 1], refs[i][0]) for i in range(B)])
    t = torch.tensor([O.translation_error(hipT[i:i + 1], refs[i][1]).max().item() for i in range(B)])
    print("# %s weights, %s (N=%d J=%d), pairs %d..%d, precision %s: %d of %d within 1e-5; R max %.2e median %.2e" % (
        args.profile, args.workload, N, J, first, first + B - 1, args.precision, int(((r < 1e-5) & (t < 1e-5)).sum()), B, r.max(), r.median()))
    if args.ids:
        ids = [int(v) - first for v in args.ids.split(",") if v and first <= int(v) < first + B]
    elif args.all:
        ids = list(range(B))
    else:
        ids = [int(i) for i in torch.nonzero((r >= 1e-5) | (t >= 1e-5)).flatten()]
    stages = ("emb", "ft", "f", "f>ovl", "o", "f2", "em", "em+f2", "muf", "HIP")
    print("# d(inject X): oracle downstream of the HIP path's stage X, distance of its R to the oracle's R [rad]")
    print("#  pair  " + " ".join("%9s" % s for s in stages))
    rows = {}
    for i in ids:
        s1, t1, st1 = src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1]
        refR = refs[i][0]
        cap = caps[i]
        knn = {k: cap[k] for k in ("knn_idx_src", "knn_idx_tgt")}

        def run(inj):
            with torch.no_grad():
                return O.forward(P, cfg, s1, t1, st1, inject=dict(knn, **inj))[0]
        em = {"em_" + s: (g["gamma"][c][None], g["pi"][c][None], g["mu"][c][None]) for s, c in (("src", i), ("tgt", B + i))}
        f2 = dict(zip(("f2_src", "f2_tgt"), feat("f2", i)))
        o = {"o_src": g["o"][i][None], "o_tgt": g["o"][B + i][None]}
        fs, ft_ = feat("f", i)
        row = []
        for key in ("emb", "ft", "f"):
            a, b = feat(key, i)
            row.append(d_rot(run({key + "_src": a, key + "_tgt": b}), refR))
        # f into the overlap chain only: HIP's f produces o in the oracle's arithmetic, the oracle's own f feeds the last transformer
        with torch.no_grad():
            capo = {}
            O.forward(P, cfg, s1, t1, st1, cap=capo, inject=dict(knn, f_src=fs, f_tgt=ft_))
        row.append(d_rot(run({"o_src": capo["o_src"], "o_tgt": capo["o_tgt"]}), refR))
        row.append(d_rot(run(o), refR))
        row.append(d_rot(run(f2), refR))
        row.append(d_rot(run(em), refR))
        row.append(d_rot(run(dict(em, **f2)), refR))
        row.append(d_rot(run(dict(em, **f2, muf_src=g["muf"][i][None], muf_tgt=g["muf"][B + i][None])), refR))
        row.append(r[i].item())
        rows[i] = row
        print("  %5d  " % (first + i) + " ".join("%9.2e" % v for v in row))

    # ---- accuracy per stage against the fp64 evaluation, and single kernels on the oracle's inputs
    print("\n# per stage: rms |HIP - f64| / rms |reference fp32 - f64|, both relative to the stage's rms   (same kNN graph; > 1: this path is less accurate than the reference's fp32 there)")
    print("#  pair  " + " ".join("%17s" % s for s in ("emb", "ft", "f", "o", "f2", "gamma", "mu", "muf")) + " | single kernels on the oracle's inputs: E/M (mu), feature means, match+solve (R)")
    dev = "cuda:0"
    for i in ids:
        s1, t1, st1 = src[i:i + 1], tgt[i:i + 1], starts[:, i:i + 1]
        cap = caps[i]
        knn = {k: cap[k] for k in ("knn_idx_src", "knn_idx_tgt")}
        with torch.no_grad():
            c64 = {}
            O.forward(P64, cfg, s1.double(), t1.double(), st1, cap=c64, inject=knn)
        cells = []
        for key, hip in (("emb", feat("emb", i)), ("ft", feat("ft", i)), ("f", feat("f", i)), ("o", (g["o"][i][None], g["o"][B + i][None])), ("f2", feat("f2", i)),
                         ("gamma", (g["gamma"][i][None], g["gamma"][B + i][None])), ("mu", (g["mu"][i][None], g["mu"][B + i][None])),
                         ("muf", (g["muf"][i][None], g["muf"][B + i][None]))):
            # (root-mean-square over the map: the maximum is one outlier element; relative to the map's own rms)
            nrm = max(c64[key + "_src"].pow(2).mean().sqrt().item(), 1e-30)
            dh = max((hip[n].double() - c64[key + "_" + s]).pow(2).mean().sqrt().item() for n, s in enumerate(("src", "tgt"))) / nrm
            dr = max((cap[key + "_" + s].double() - c64[key + "_" + s]).pow(2).mean().sqrt().item() for s in ("src", "tgt")) / nrm
            cells.append("%8.1e/%8.1e" % (dh, dr))
        # single kernels on the oracle's fp32 inputs
        with torch.no_grad():
            xyz = torch.cat([s1.transpose(1, 2), t1.transpose(1, 2)]).contiguous().to(dev)
            o_ref = torch.cat([cap["o_src"], cap["o_tgt"]]).contiguous().to(dev)
            idsj = torch.cat([cap["fpsJ_src"], cap["fpsJ_tgt"]]).to(torch.int32).contiguous().to(dev)
            gam, pi, mu = ops.gmm_em(xyz, o_ref, idsj, iters=10, sk_iters=10, epsilon=1e-2, tau=1.0, thresh=1e-2, group_size=1)[:3]
            mu_ref = torch.cat([cap["mu_src"], cap["mu_tgt"]])
            d_em = (mu.cpu() - mu_ref).abs().max().item()
            d_em_R = d_rot(O.forward(P, cfg, s1, t1, st1, inject=dict(knn, em_src=(gam[0:1].cpu(), pi[0:1].cpu(), mu[0:1].cpu()),
                                                                em_tgt=(gam[1:2].cpu(), pi[1:2].cpu(), mu[1:2].cpu())))[0], refs[i][0])
            g_ref = torch.cat([cap["gamma_src"], cap["gamma_tgt"]]).contiguous().to(dev)
            pi_ref = torch.cat([cap["pi_src"], cap["pi_tgt"]]).contiguous().to(dev)
            f2_ref = torch.cat([cap["f2_src"], cap["f2_tgt"]]).transpose(1, 2).reshape(2 * N, D).contiguous().to(dev)
            muf = ops.gmm_feat_mean(g_ref, pi_ref, f2_ref, 2, N)
            muf_ref = torch.cat([cap["muf_src"], cap["muf_tgt"]])
            d_muf = (muf.cpu() - muf_ref).abs().max().item()
            d_muf_R = d_rot(O.forward(P, cfg, s1, t1, st1, inject=dict(knn, muf_src=muf[0:1].cpu(), muf_tgt=muf[1:2].cpu()))[0], refs[i][0])
            Rk, tk = ops.match_kabsch(mu_ref[0:1].contiguous().to(dev), mu_ref[1:2].contiguous().to(dev), muf_ref[0:1].contiguous().to(dev), muf_ref[1:2].contiguous().to(dev), 0.05)
            d_match = d_rot(Rk.cpu(), refs[i][0])
        print("  %5d  " % (first + i) + " ".join(cells) + " | E/M mu %.1e -> R %.2e; means %.1e -> R %.2e; match+solve R %.2e" % (d_em, d_em_R, d_muf, d_muf_R, d_match))
    if rows:
        m = torch.tensor([rows[i] for i in rows])
        print("\n# median over the %d listed pairs: " % len(rows) + " ".join("%s %.2e" % (s, v) for s, v in zip(stages, m.median(0)[0].tolist())))


if __name__ == "__main__":
    main()
