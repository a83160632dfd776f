"""torch.profiler view of one training step: device time per aten / autograd operator (finds the torch-op glue that is left)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from argparse import Namespace
from torch.profiler import profile, ProfilerActivity
from ogmm_amd import synth
from ogmm_amd.gmmreg import GMMReg
from ogmm_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
cfg = Namespace(gnn_k=20, num_heads=4, km_clusters=128, overlap_radius=0.035)
model = GMMReg(512, 16, cfg); synth.fill_state_dict(model.state_dict()); model = model.to(dev)
batch = [t.to(dev) for t in synth.make_train_batch(0, B, 1024)]
starts = synth.fps_starts_for(0, B, 1024)
tr = Trainer(model)
for _ in range(2):
    tr.step(*batch, fps_starts=starts)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(*batch, fps_starts=starts)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
if len(sys.argv) > 2 and sys.argv[2] == "shapes":
    # second view: the aten glue by input shape
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof2:
        tr.step(*batch, fps_starts=starts)
        torch.cuda.synchronize()
    rows = [e for e in prof2.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
    rows.sort(key=lambda e: -e.self_device_time_total)
    for e in rows[:60]:
        print("%-28s x%-4d %9.3f ms  %s" % (e.key, e.count, e.self_device_time_total / 1e3, str(e.input_shapes)[:150]))
