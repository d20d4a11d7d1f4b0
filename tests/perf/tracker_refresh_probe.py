import sys, os, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import slam_synthetic as ss
dev = torch.device("cuda", 0)
seq = ss.SyntheticSequence(4, 240, 320, device=dev, seed=0)
ops = ss.ProductOps(seq, dev, seed=0, fused=True)
ops.update_tracker_copy()
def T(f, n=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    h = time.perf_counter() - t
    torch.cuda.synchronize()
    return 1e3 * h / n, 1e3 * (time.perf_counter() - t) / n
dst = list(ops.c_track.values()) + [m.flat_params() for m in ops.decoders_track.children()]
src = list(ops.c.values()) + [m.flat_params() for m in ops.decoders.children()]
print("sizes", [tuple(t.shape) for t in dst], [t.is_contiguous() for t in dst])
print("flat_params x8 host/total ms", T(lambda: [m.flat_params() for m in ops.decoders_track.children()] + [m.flat_params() for m in ops.decoders.children()]))
with torch.no_grad():
    print("foreach_copy", T(lambda: torch._foreach_copy_(dst, src)))
    print("per-tensor copy", T(lambda: [d.copy_(s) for d, s in zip(dst, src)]))
    def rp():
        torch._foreach_copy_(dst, src); ops.decoders_track.repack()
    print("copy+repack", T(rp))
    def full():
        ops.map_version = getattr(ops, "map_version", 0) + 1
        ops.update_tracker_copy()
    print("update_tracker_copy", T(full))
    flat_d = [t.view(-1) if t.is_contiguous() else t.permute(0, 2, 3, 4, 1).reshape(-1) for t in dst]
    print("views share", [f.data_ptr() == t.data_ptr() for f, t in zip(flat_d, dst)])
