"""Measurement tooling (not a test, not collected by pytest): the minimal reproducer behind the warning in nice_slam_amd/graphs.py.
On ROCm 7.2 / torch 2.10 ``hipStreamEndCapture`` dies with SIGSEGV -- instead of reporting unjoined work -- when a captured backward
reaches an AccumulateGrad node that an EAGER iteration created on another stream (a leaf pose tensor whose loss tensor is still alive).
    python tests/perf/capture_segv_probe.py <variant>
<variant> is a string of flags, tested by substring: nocam | onecam | plaincam (which pose leaves feed the rays; default: two cameras
through get_camera_from_tensor), allzero, nostep, deepcopy, ref (a torch.optim.Adam twin stepped alongside), float (print the losses),
captured (wrap the iteration in graphs.CapturedStep instead of running it eagerly).  Each flag narrows which leaf drags the eager
stream into the capture.  Runs on the GPU box only; a variant either finishes or dies (faulthandler shows where)."""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, faulthandler
faulthandler.enable()
from scene_util import build_product, make_scene
import nice_slam_amd as nsa
variant = sys.argv[1]
DEV = "cuda:0"
sc = make_scene(seed=12, n_rays=256, small=True)
renderer, decA, grids = build_product(sc, DEV)
camA = torch.tensor([[1.0, 0.02, -0.01, 0.03, 0.1, -0.2, 0.05], [0.98, -0.03, 0.02, 0.01, -0.1, 0.1, 0.0]], device=DEV).requires_grad_(True)
rays_o = sc["rays_o"].to(DEV); rays_d = sc["rays_d"].to(DEV); gt = sc["gt_depth"].to(DEV)
tgt = torch.rand(rays_o.shape[0], 3).to(DEV)
def loss_of(dec, cam):
    if "nocam" in variant:
        o = rays_o
    elif "onecam" in variant:
        o = rays_o + nsa.get_camera_from_tensor(cam)[0, :3, 3] * 1e-2
    elif "plaincam" in variant:
        o = rays_o + cam[0, 4:] * 1e-2
    else:
        o = rays_o + nsa.get_camera_from_tensor(cam)[0, :3, 3] * 1e-2 + nsa.get_camera_from_tensor(cam)[1, :3, 3] * 1e-2
    d, u, c = renderer.render_batch_ray(grids, dec, rays_d, o, DEV, "color", gt_depth=gt)
    return (d - gt).abs().sum() + 0.2 * (c - tgt).abs().sum()
ents = [decA.color_decoder] + ([] if "nocam" in variant else [camA])
flat = nsa.FlatAdam(ents, lr=1e-3)
def it():
    flat.zero_grad(set_to_none=True)
    if "allzero" in variant:
        for p in decA.parameters(): p.grad = None
    loss = loss_of(decA, camA)
    loss.backward()
    if "nostep" not in variant:
        flat.step()
    return loss
if "deepcopy" in variant or "ref" in variant:
    decB = copy.deepcopy(decA)
if "ref" in variant:
    camB = camA.detach().clone().requires_grad_(True)
    ref = torch.optim.Adam([{"params": list(decB.color_decoder.parameters()), "lr": 1e-3}, {"params": [camB], "lr": 1e-3}])
for _ in range(3):
    la = it()
    if "ref" in variant:
        ref.zero_grad(set_to_none=True); lb = loss_of(decB, camB); lb.backward(); ref.step()
    if "float" in variant:
        print(float(la))
torch.cuda.synchronize()
buf = torch.zeros(1, device=DEV)
if "captured" in variant:
    step = nsa.graphs.CapturedStep(it)
    for _ in range(3): step()
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        buf.copy_(it().detach().reshape(1).float())
    for _ in range(3): g.replay()
torch.cuda.synchronize()
print("OK", variant, float(buf))
