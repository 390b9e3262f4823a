"""RoIAlign forward at the bench's shape (512 rois from the model itself, 38x63x1024 map inside the 2048-wide buffer)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dana_amd
from dana_amd import ops, synthetic as S
dev = torch.device("cuda:0")
m = dana_amd.get_model("DAnA", pretrained=False, use_BA_block=True, way=2, shot=3, classes=["fg", "bg"])
m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=11, profile="test"))
m.to(dev).train()
inputs = [t.to(dev) for t in S.episode_inputs(4, 2, 3, 600, 1000, seed=1996)]
m._capture = {}
np.random.seed(1)
with torch.no_grad():
    out = m(*inputs)
corr, B, fh, fw = m._capture["corr"]
rois = out[0].reshape(-1, 5).contiguous()
plan = m._get_plan()
f = lambda: ops.roi_align_forward_nhwc(corr, B, fh, fw, 1024, 2048, rois, 1.0 / 16.0, 7, 0, pe=plan["pe49"])  # noqa: E731
for _ in range(5):
    f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
alg = 4.0 * (rois.size(0) * 1024 * 49 * 2 + B * 1024 * fh * fw + 5 * rois.size(0))
print("roi_align_fwd_nhwc: %d rois, %.1f us, algorithmic %.0f MB -> %.2f TB/s" % (rois.size(0), us, alg / 1e6, alg / us / 1e6))
# What binds it: every sample of every bin is four taps, each a 1 KB wave-level load (64 lanes x 16 B) per 256 channels, and a
# CU's vector-memory path returns 64 B per clock. Count the taps the rois ask for (ROIAlign_cuda.cu:78-103: adaptive grid =
# ceil(roi / 7) per axis) and price them at that rate -- the floor of ANY kernel that fetches the four taps of every sample.
r = rois.cpu().numpy().astype(np.float32)
rw = np.maximum(r[:, 3] / np.float32(16) - r[:, 1] / np.float32(16), 1.0)
rh = np.maximum(r[:, 4] / np.float32(16) - r[:, 2] / np.float32(16), 1.0)
gh, gw = np.ceil(rh / 7.0), np.ceil(rw / 7.0)
samples = float((gh * gw).sum() * 49)
tap_bytes = samples * 4 * 1024 * 4
props = torch.cuda.get_device_properties(0)
cus, clk = props.multi_processor_count, float(getattr(props, "clock_rate", 2400000)) * 1e3  # (kHz; MI355X: 2.4 GHz)
floor_us = tap_bytes / (cus * 64.0 * clk) * 1e6
distinct = float(((gh + 1) * (gw + 1)).sum() * 49)
print("samples per bin: mean %.2f (grid %.2f x %.2f), %.2f M samples -> %.2f GB of tap loads through the CUs' vector-memory "
      "path; at %d CUs x 64 B/clk x %.2f GHz that is %.1f us (measured %.1f us = %.0f %% of it)"
      % (samples / 49 / len(r), gh.mean(), gw.mean(), samples / 1e6, tap_bytes / 1e9, cus, clk / 1e9, floor_us, us,
         100.0 * floor_us / us))
print("distinct cells per bin (every tap shared between neighbouring samples fetched once): %.2f M cell loads = %.0f %% of the taps"
      % (distinct / 1e6, 100.0 * distinct / (4 * samples)))
# Is it the work's SHAPE? per-roi sample counts, their share per XCD run of the contiguous mapping, and the same launch on
# (a) the rois sorted by falling sample count (long bins first), (b) every roi shrunk to one sample per bin
per_roi = (gh * gw)
order = np.argsort(-per_roi, kind="stable")
print("samples per bin by roi: max %d, p99 %d, p90 %d, median %d; the 16 largest rois hold %.0f %% of all samples"
      % (per_roi.max(), np.percentile(per_roi, 99), np.percentile(per_roi, 90), np.median(per_roi),
         100.0 * per_roi[order[:16]].sum() / per_roi.sum()))
chunks = np.array_split(per_roi, 8)
print("share of the samples per XCD run (8 contiguous runs of rois): %s" % " ".join("%.0f%%" % (100.0 * c.sum() / per_roi.sum()) for c in chunks))


def timed(rr):
    ff = lambda: ops.roi_align_forward_nhwc(corr, B, fh, fw, 1024, 2048, rr, 1.0 / 16.0, 7, 0, pe=plan["pe49"])  # noqa: E731
    for _ in range(5):
        ff()
    e0.record()
    for _ in range(50):
        ff()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 50


print("as sampled: %.1f us; sorted by falling sample count: %.1f us" % (timed(rois), timed(rois[torch.from_numpy(order.copy()).to(rois.device)].contiguous())))
small = rois.clone()
small[:, 3] = torch.minimum(small[:, 3], small[:, 1] + 100.0)
small[:, 4] = torch.minimum(small[:, 4], small[:, 2] + 100.0)
print("every roi at most 100 x 100 px (one sample per bin): %.1f us" % timed(small))
