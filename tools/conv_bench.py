"""Per-layer conv micro-benchmark (optimisation harness): times fwd / dgrad / wgrad of the net's distinct layer
shapes at a given batch, interleaving kernel variants in one process (HIP events), prints TFLOP/s."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops

LAYERS = [  # name, cin, cout, k, s, d, H (input), count in net
    ("b2.2b1 128->128 3x3 @112", 128, 128, 3, 1, 1, 112, 5),
    ("b3.2b1 256->256 3x3 @56", 256, 256, 3, 1, 1, 56, 5),
    ("b4.2a 256->512 3x3 s2", 256, 512, 3, 2, 1, 56, 1),
    ("b4.2b1 512->512 3x3 @28", 512, 512, 3, 1, 1, 28, 12),
    ("b5.2b1 512->1024 3x3 d2", 512, 1024, 3, 1, 2, 28, 3),
    ("b5_1.2a 1024->512 3x3 d2", 1024, 512, 3, 1, 2, 28, 2),
    ("b6.2b1 512->1024 3x3 d4", 512, 1024, 3, 1, 4, 28, 1),
    ("b7.2b1 1024->2048 3x3 d4", 1024, 2048, 3, 1, 4, 28, 1),
    ("b7.b1 2048->4096 1x1", 2048, 4096, 1, 1, 1, 28, 2),
    ("b7.2a 2048->1024 1x1", 2048, 1024, 1, 1, 1, 28, 1),
    ("b6.b1 1024->2048 1x1", 1024, 2048, 1, 1, 1, 28, 2),
    ("b7 fused 4096->4096 1x1", 4096, 4096, 1, 1, 1, 28, 1),
    ("b7 fused dgrad 5120->2048", 2048, 5120, 1, 1, 1, 28, 1),
    ("b5.b1 512->1024 1x1", 512, 1024, 1, 1, 1, 28, 1),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--variants", default="3stage=0,3stage=1")
    ap.add_argument("--layers", default="")
    ap.add_argument("--halo-ring", type=int, default=3)
    ap.add_argument("--data", default="randn", choices=["randn", "relu", "zeros"],
                    help="operand values: dense random | max(randn, 0) activations | all zeros.  The big kernels sit at the 1400 W package cap on random data "
                         "(NOTES 7.28): a sustained loop then measures energy per FLOP; --data zeros (2.4 GHz, ~950 W) measures CYCLES per FLOP")
    ap.add_argument("--tile", type=int, default=224, choices=[224, 256], help="input tile edge: the table's maps are for 224 (28 / 56 / 112 wide); 256 scales them to 32 / 64 / 128")
    ap.add_argument("--epi", default="raw", help="raw: store only | full: fwd = +residual -> raw + BN/ReLU out, dgrad = ReLU mask + add1 -> out")
    args = ap.parse_args()
    lib = _lib.use_debug_library()  # the ps_debug_* switches live in libpistoseg_hip_debug.so only
    D = torch.device("cuda:0")
    dt = torch.bfloat16
    variants = [v.split("=") for v in args.variants.split(",")]
    sel = [int(i) for i in args.layers.split(",")] if args.layers else range(len(LAYERS))
    for li in sel:
        name, cin, cout, k, s, d, H, cnt = LAYERS[li]
        H = H * args.tile // 224
        n = args.batch
        spec = ops.ConvSpec(cin, cout, k, s, d)
        ho, wo = spec.out_hw(H, H)
        rnd = {"randn": lambda *sh: torch.randn(*sh, device=D), "relu": lambda *sh: torch.randn(*sh, device=D).relu(),
               "zeros": lambda *sh: torch.zeros(*sh, device=D)}[args.data]
        wscale = 0.0 if args.data == "zeros" else 0.02
        x = rnd(n, H, H, cin).to(dt)
        wf = (torch.randn(cout, k, k, cin, device=D) * wscale).to(dt)
        wd = (torch.randn(cin, k, k, cout, device=D) * wscale).to(dt)
        gy = (torch.randn(n, ho, wo, cout, device=D) * (0.0 if args.data == "zeros" else 1.0)).to(dt)
        y = torch.empty(n, ho, wo, cout, device=D, dtype=dt)
        gx = torch.empty(n, H, H, cin, device=D, dtype=dt)
        dw = torch.zeros(cout, k, k, cin, device=D)
        flops = 2.0 * n * ho * wo * cout * cin * k * k
        if args.epi == "full":
            res_in = torch.randn(n, ho, wo, cout, device=D).to(dt)
            y2 = torch.empty_like(y)
            bsc, bsh = torch.rand(cout, device=D) + 0.5, torch.randn(cout, device=D)
            gadd = torch.randn(n, H, H, cin, device=D).to(dt)
            isc = torch.rand(cin, device=D) + 0.5
            f_fwd = lambda: ops.conv2d_fwd(spec, x, wf, add0=res_in, out_raw=y, bn_scale=bsc, bn_shift=bsh, out_act=y2)
            f_dg = lambda: ops.conv2d_dgrad(spec, gy, wd, (H, H), mask_src=x, bn_scale=isc, add1=gadd, out=gx)
        else:
            f_fwd = lambda: ops.conv2d_fwd(spec, x, wf, out_raw=y)
            f_dg = lambda: ops.conv2d_dgrad(spec, gy, wd, (H, H), out_raw=gx)
        fns = {"fwd": f_fwd, "dgrad": f_dg,
               "wgrad": lambda: ops.conv2d_wgrad(spec, x, gy, dw)}
        line = f"{name:28s}"
        for what in args.what.split(","):
            res = {}
            for r in range(3):  # interleaved rounds
                for vn, vv in variants:
                    lib.ps_debug_reset()
                    lib.ps_debug_set_halo_ring(args.halo_ring)
                    getattr(lib, "ps_debug_set_" + vn)(int(vv))
                    fns[what]()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        fns[what]()
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault((vn, vv), []).append(e0.elapsed_time(e1) / args.iters)
            line += f" | {what}:"
            for (vn, vv), ts in res.items():
                t = min(ts)
                line += f" {vn}={vv}: {t*1e3:7.1f}us {flops/t/1e9:6.0f}TF"
        print(line, flush=True)
    lib.ps_debug_reset()

if __name__ == "__main__":
    main()
