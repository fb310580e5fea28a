"""Measurement of the SURVEY 8f rows on one MI355X (synthetic tiles; HIP events on the current stream; CPU oracle timed beside on a
bounded sample).  Prints one JSON line per row:
  sliding : softmax -> f64 canvas scatter-add of 256x256 3-class tiles + final resize/argmax/confusion, tiles/s and GB/s
            against the HBM roofline (algorithmic bytes/tile = read 3*S*S*4 + f64 read-modify-write of 3+1 canvas planes)
  tta     : stage-2 inference with the d4 wrapper (8 views batched into one forward), tiles/s
  oeem    : stage-0 multi-scale CAM of one 1024x1024 image (5 scales, 224-crops on a stride-56 grid...), crops/s
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def ev_time(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="sliding,tta,oeem")
    ap.add_argument("--cpu", action="store_true", help="time the CPU oracle on a bounded sample beside each row")
    args = ap.parse_args()
    D = torch.device("cuda:0")
    from oracle import ref_cpu
    from pistoseg_amd.seg_model import ResNet38dSeg
    from pistoseg_amd.trainer import init_weights_he

    if "sliding" in args.what:
        from pistoseg_amd.sliding import SlidingWindowAccumulator
        c, s, n = 3, 256, 64
        w, h = 4096, 4096
        sizes = {"1": (w, h)}
        pos = [(y, x) for y in range(0, h - s + 1, 128) for x in range(0, w - s + 1, 128)][:n * 8]
        names = [f"1_1.0_{y}_{x}-x.png" for y, x in pos]
        logits = torch.randn(n, c, s, s, device=D)
        acc = SlidingWindowAccumulator(c, D, lambda idx: sizes[idx])
        batches = [names[i:i + n] for i in range(0, len(names), n)]
        def run():
            for nm in batches:
                acc.add_batch(logits[:len(nm)], nm, [s] * len(nm), [s] * len(nm))
        dt = ev_time(run, 3)
        tiles = len(names)
        bytes_per_tile = c * s * s * 4 + (c + 1) * s * s * 8 * 2
        gt = torch.randint(0, 4, (h, w), dtype=torch.uint8)
        t0 = time.perf_counter(); acc.big_mask_iou(lambda idx: gt).confusion_matrix; torch.cuda.synchronize(); t_fin = time.perf_counter() - t0
        out = {"row": "sliding", "tiles_per_s": round(tiles / dt, 1), "achieved_GBps": round(tiles * bytes_per_tile / dt / 1e9, 1), "peak_GBps": 8000,
               "frac": round(tiles * bytes_per_tile / dt / 8e12, 4), "finalize_4096x4096_ms": round(t_fin * 1e3, 2), "tile": s, "classes": c}
        if args.cpu:
            lg = logits[:16].cpu(); nm = names[:16]
            t0 = time.perf_counter()
            ref_cpu.sliding_window_big_masks([(lg, nm, [s] * 16, [s] * 16)], sizes, c)
            out["cpu_tiles_per_s_incl_finalize"] = round(16 / (time.perf_counter() - t0), 2)
            out["cpu_cores"] = torch.get_num_threads()
        print(json.dumps(out), flush=True)

    if "tta" in args.what:
        from pistoseg_amd.tta import SegmentationTTAWrapper
        model = ResNet38dSeg(classes=3, precision="bf16"); init_weights_he(model, 42); model = model.to(D); model.eval()
        x = torch.randn(32, 3, 224, 224, device=D)
        wrap = SegmentationTTAWrapper(model)
        with torch.no_grad():
            dt1 = ev_time(lambda: model(x), 5)
            dt8 = ev_time(lambda: wrap(x), 3)
        print(json.dumps({"row": "tta", "plain_tiles_per_s": round(32 / dt1, 1), "d4_tta_tiles_per_s": round(32 / dt8, 1),
                          "views_per_s": round(8 * 32 / dt8, 1), "batch": 32, "dtype": "bf16"}), flush=True)

    if "oeem" in args.what:
        from pistoseg_amd.oeem import image_cam_32x32, wideResNet
        net = wideResNet(num_class=3, precision="bf16"); init_weights_he(net, 42); net = net.to(D); net.eval()
        w = h = 1024; side = 224; scales = [1.0, 1.25, 1.5, 1.75, 2.0]
        rs = np.random.RandomState(0)
        ims, poss = [], []
        for sc in scales:
            w_, h_ = int(w * sc), int(h * sc)
            ys = sorted(set(list(range(0, w_ - side + 1, side // 3 * 2)) + [w_ - side]))
            xs = sorted(set(list(range(0, h_ - side + 1, side // 3 * 2)) + [h_ - side]))
            pos = [(y, x) for y in ys for x in xs]
            ims.append(torch.randn(len(pos), 3, side, side)); poss.append(pos)
        ncrops = sum(len(p) for p in poss)
        ims_d = [t.to(D) for t in ims]
        dt = ev_time(lambda: image_cam_32x32(net, ims_d, poss, scales, (w, h), side, batch_size=64), 2)
        print(json.dumps({"row": "oeem", "crops_per_image": ncrops, "image_ms": round(dt * 1e3, 1), "crops_per_s": round(ncrops / dt, 1),
                          "image": f"{w}x{h}", "scales": scales, "dtype": "bf16"}), flush=True)


if __name__ == "__main__":
    main()
