"""A short training run on fresh synthetic batches every step (ogmm_amd.trainer.Trainer): loss parts and registration errors over
time, skipped steps, throughput.  usage: train_run.py [steps] [pairs_per_step]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
from ogmm_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
device_data = len(sys.argv) > 3 and sys.argv[3] == "device"      # samples made on the GPU by ogmm_amd.augment (the reference's crop chain, 717 points)
N, J = 1024, 16
dev = "cuda:0"
torch.manual_seed(0)
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
model = GMMReg(512, J, cfg).to(dev)            # PyTorch default initialisation, as the reference's train.py starts from
tr = Trainer(model, lr=1e-4)
if device_data:
    import numpy as np
    from ogmm_amd import augment
    pool = torch.stack([torch.from_numpy(synth._patch_cloud(np.random.Generator(np.random.PCG64(500 + i)), 1024)).float() for i in range(256)]).to(dev)
    gen = torch.Generator(device=dev).manual_seed(1)
t0 = time.perf_counter()
for it in range(steps):
    if device_data:
        shapes = pool[torch.randint(0, pool.shape[0], (B,), generator=gen, device=dev)]
        smp = augment.crop_pipeline(shapes, augment.draw(B, 1024, 717, gen, dev), n_out=717)
        batch = [smp["src_xyz"].transpose(1, 2).contiguous(), smp["tgt_xyz"].transpose(1, 2).contiguous(), smp["transform_gt"],
                 smp["src_overlap"], smp["tgt_overlap"]]
    else:
        batch = [t.to(dev) for t in synth.make_train_batch(10000 + it * B, B, N, "partial")]
    info = tr.step(*batch)
    if it % 5 == 0 or it == steps - 1:
        p = {k: float(v) for k, v in info["parts"].items()}
        print("step %3d loss %.4f (dcp %.4f clu %.4f mse %.4f welsch %.3f) r_err %.2f deg t_err %.3f scale %g skipped %d" % (
            it, float(info["loss"]), p["dcp"], p["clu"], p["mse"], p["welsch"], float(info["r_err_deg"]), float(info["t_err"]), tr.loss_scale, tr.skipped_steps), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d steps of %d pairs in %.1f s (%.0f pairs/s incl. %s batch synthesis)" % (steps, B, dt, steps * B / dt, "device-side" if device_data else "host-side"))
