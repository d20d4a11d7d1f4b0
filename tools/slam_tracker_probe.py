"""Diagnostic for tools/slam_synthetic.py: map frame 0 with its ground-truth pose, then start the tracker of the SAME frame
from a perturbed pose and print the pose error per iteration (does the pose gradient of the render path pull it back?)."""
import copy
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import slam_synthetic as ss


def main():
    dev = torch.device("cuda", 0)
    iters_first = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    pixels = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    cfg = copy.deepcopy(ss.DEFAULT_CFG)
    cfg["mapping"]["iters_first"] = iters_first
    cfg["tracking"].update(lr=lr, pixels=pixels)
    torch.manual_seed(0)
    seq = ss.SyntheticSequence(4, 240, 320, device=dev)
    ops = ss.ProductOps(seq, dev)
    slam = ss.MiniSLAM(ops, seq, cfg)
    color, depth, gt = seq.frame(0)
    slam.est[0] = gt.clone()
    slam.optimize_map(iters_first, cfg["mapping"]["lr_first_factor"], 0, color, depth, gt)
    print("map loss", slam.last_map_loss)
    from scipy.spatial.transform import Rotation
    for trans_cm, rot_deg in ((2.0, 0.0), (0.0, 1.0), (2.0, 1.0)):
        pert = gt.clone()
        pert[:3, 3] += torch.tensor([trans_cm / 100 / np.sqrt(3)] * 3, device=dev)
        dR = torch.tensor(Rotation.from_euler("y", rot_deg, degrees=True).as_matrix(), dtype=torch.float32, device=dev)
        pert[:3, :3] = dR @ pert[:3, :3]
        cam = ss.get_tensor_from_camera(pert).requires_grad_(True)
        opt = torch.optim.Adam([cam], lr=lr)
        gt_t = ss.get_tensor_from_camera(gt)
        line = []
        for it in range(60):
            loss = slam._track_iter(cam, color, depth, opt)
            with torch.no_grad():
                c2w = ss.get_camera_from_tensor(cam)
                te = float((c2w[:3, 3] - gt[:3, 3]).norm()) * 100
                re = float(torch.rad2deg(torch.acos(((c2w[:3, :3].T @ gt[:3, :3]).trace().clamp(-1, 3) - 1) / 2)))
            if it % 5 == 0 or it == 59:
                line.append(f"{it}:{loss:.0f}/{te:.2f}cm/{re:.2f}deg")
        print(f"perturb {trans_cm}cm {rot_deg}deg ->", " ".join(line))


if __name__ == "__main__":
    main()
