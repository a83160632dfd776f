"""Generates tests/golden/deepgmr_b2_n512_j16.npz by running the reference's DeepGMR baseline (baseline/deepgmr.py) on CPU in eval
mode with the closed-form weights of ogmm_amd/synth.py.  `gmm_register` hard-codes `.cuda()` (baseline/deepgmr.py:30-31);
`Tensor.cuda` is patched to the identity for this run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ref_harness import default_config, import_reference  # noqa: E402
from ogmm_amd import synth                                        # noqa: E402


def main():
    import_reference()
    import baseline.deepgmr as ref
    torch.set_num_threads(8)
    B, N, J = 2, 512, 16
    cfg = default_config(n_clusters=J)
    net = ref.DeepGMR(512, J, cfg).eval()
    synth.fill_state_dict(net.state_dict())
    # With the plain fill the J cluster scores are nearly uniform, the 3x3 matrix of gmm_register is ~1e-6 and the rotation is set by
    # the +1e-4 the reference adds to every entry (an ill-conditioned test).  Sharper cluster logits give the matrix real content.
    C6_SCALE = 60.0
    with torch.no_grad():
        net.state_dict()["cluster.net.6.weight"].mul_(C6_SCALE)
    src, tgt, R, t = synth.make_batch(800, B, N, "partial")
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            rot, second = net(src, tgt)
    finally:
        torch.Tensor.cuda = real_cuda
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deepgmr_b2_n512_j16.npz")
    np.savez_compressed(path, src=src.numpy(), tgt=tgt.numpy(), R_gt=R.numpy(), t_gt=t.numpy(), R=rot.numpy(), second=second.numpy(),
                        keys=np.array(sorted(net.state_dict().keys())), c6_scale=np.float32(C6_SCALE))
    # conditioning report: singular values of the registration matrix, recomputed with the reference's own pieces
    with torch.no_grad():
        from lib.utils import gmm_params
        fs, ft = net.backbone(src), net.backbone(tgt)
        gs, gt_ = torch.softmax(net.cluster(fs), dim=1), torch.softmax(net.cluster(ft), dim=1)
        pi_s, mu_s, _ = gmm_params(gs.transpose(-1, -2), src.transpose(-1, -2), True)
        _, mu_t, sg_t = gmm_params(gt_.transpose(-1, -2), tgt.transpose(-1, -2), True)
        c_s, c_t = pi_s.unsqueeze(1) @ mu_s, pi_s.unsqueeze(1) @ mu_t
        Ms = torch.sum((pi_s.unsqueeze(2) * (mu_s - c_s)).unsqueeze(3) @ (mu_t - c_t).unsqueeze(2) @ sg_t.inverse(), dim=1)
        print("singular values of Ms:", torch.linalg.svdvals(Ms), "max gamma:", gs.max().item())
    print(path, rot[0], second)


if __name__ == "__main__":
    main()
