"""TEST / MEASUREMENT INFRASTRUCTURE: which ATen / nsr launches one fused mapping (or tracking) iteration issues, with the Python
frame that caused each (torch.profiler, eager).      python tests/perf/launch_trace.py [color|middle|fine|tracking]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import nice_slam_amd as nsa
from torch.profiler import profile, ProfilerActivity

what = sys.argv[1] if len(sys.argv) > 1 else "color"
dev = torch.device("cuda", 0)
cfg = "tracking" if what == "tracking" else "1"
sc = bench.build_scene(cfg, dev)
renderer, dec = sc["renderer"], sc["dec"]
H, W, fx, fy, cx, cy = sc["intr"]
frames = [(c.to(dev), d.to(dev), col.to(dev)) for c, d, col in sc["frames"]]
if what == "tracking":
    grids = {k: v.detach() for k, v in sc["grids"].items()}
    for p in dec.parameters():
        p.requires_grad_(False)
    cam = frames[0][0][:3].clone().requires_grad_(True)
else:
    grids = {k: v.requires_grad_(True) for k, v in sc["grids"].items()}
    for p in dec.parameters():
        p.requires_grad_(True)


def step():
    for g in grids.values():
        g.grad = None
    for p in dec.parameters():
        p.grad = None
    if what == "tracking":
        cam.grad = None
        nsa.tracking_loss(renderer, grids, dec, cam, frames[0][1], frames[0][2], 200, 100, 100, w_color=0.5).backward()
    else:
        nsa.mapping_loss(renderer, grids, dec, frames, 200, what, w_color=0.2).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA or (e.cuda_time_total > 0 and not e.cpu_children)]
print("kernels of one iteration (%s):" % what)
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        print("  %-90s %7.1f us" % (e.name[:90], e.cuda_time_total if hasattr(e, "cuda_time_total") else 0.0))
print()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60, max_src_column_width=110))
