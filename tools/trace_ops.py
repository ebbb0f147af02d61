"""One benchmark step under torch.profiler, grouped by op + input shapes: finds where the elementwise /
copy time of the step comes from (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dreammat_amd
from dreammat_amd.data import RandomCameraDataModule
from dreammat_amd.system import Trainer, to_device
from torch.profiler import profile, ProfilerActivity

_argv, sys.argv = sys.argv, sys.argv[:1]
a = bench.parse()                      # the bench defaults: BASELINE configs[2]
sys.argv = _argv
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dreammat_amd._import_plugins()
torch.manual_seed(0)
lat = [bench.synthetic_latlong(i) for i in range(5)]
system = dreammat_amd.find("dreammat-system")(bench.system_config(a, 8), material_kwargs={"latlongs": lat})
system.renderer.debug_outputs = False
dm = RandomCameraDataModule(cfg={"height": 512, "width": 512, "batch_size": 8, "use_fix_views": True,
                                 "camera_distance_range": [3.0, 4.0], "fovy_range": [25, 45], "camera_perturb": 0.0,
                                 "center_perturb": 0.0, "up_perturb": 0.0, "elevation_range": [-20, 45]}, device=dev)
dm.setup("fit"); system.on_fit_start(); system.configure_optimizers()
tr = Trainer(system, dm, max_steps=10 ** 9)
batches = [to_device(dm.train_dataset.collate(), dev) for _ in range(3)]
for i in range(2):
    tr.train_one_step(batches[i])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_one_step(batches[2])
    torch.cuda.synchronize()
tab = prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=40, max_shapes_column_width=90)
open("gpurun_out/trace_ops.txt", "w").write(tab)
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0:
        rows.append((t, e.count, e.key, str(e.input_shapes)))
rows.sort(reverse=True)
with open("gpurun_out/trace_ops.csv", "w") as f:
    f.write("self_device_us,calls,name,shapes\n")
    for t, c, k, sh in rows[:500]:
        f.write(f"{t:.0f},{c},\"{k[:90]}\",\"{sh[:160]}\"\n")
print("total device us", sum(r[0] for r in rows))
