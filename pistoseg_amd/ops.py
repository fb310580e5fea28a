"""Tensor-level wrappers over the C-ABI (include/pistoseg_hip.h).

torch is used here for device memory and streams only: every function passes raw device pointers, sizes
and the current HIP stream to libpistoseg_hip.so.  Activations are channels-last `[N, H, W, C]` torch
tensors (f32 or bf16) that may be channel slices of wider buffers (`stride(2)` = channel stride `ldc`).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import PS_BF16, PS_BF16X3, PS_F16X3, PS_EPI_BNRELU, PS_EPI_NONE, PS_EPI_RELUBWD, PS_F16, PS_F32, ConvGeom, Epilogue, Tensor4

Tensor = torch.Tensor


# Optional per-launch timing of the conv kernels (bench.py's roofline leg): when PROFILE is a list, every conv
# launch is bracketed by HIP events on the current stream and appended as (kernel label, flops, start, end).
PROFILE = None


_VARIANT_NAMES = {1: "conv_igemm_kernel", 2: "conv_igemm_ws_kernel", 3: "conv_igemm_ws_kernel", 4: "conv_igemm_ws2_kernel", 5: "conv_igemm_ws2_kernel",
                  6: "conv_igemm_experimental", 7: "conv_igemm_halo_kernel", 8: "conv_gemm256_kernel"}


def _conv_label(kind: str, g: ConvGeom) -> str:
    """Kernel family serving this launch (asked of the library: ps_conv_variant mirrors the dispatcher)."""
    dt = {PS_BF16: "bf16", PS_F16: "f16", PS_BF16X3: "bf16x3", PS_F16X3: "fp16x3"}.get(g.dtype, "f32")
    if kind == "wgrad":
        v = int(_lib.load().ps_conv_wgrad_variant(C.byref(g)))
        return f"{ {2: 'conv_wgrad256_kernel', 1: 'conv_wgrad_ws2_kernel'}.get(v, 'conv_wgrad_kernel') }<{dt}>"
    v = int(_lib.load().ps_conv_variant(C.byref(g), 1 if kind == "dgrad" else 0))
    return f"{_VARIANT_NAMES.get(v, 'conv_igemm_kernel')}<{dt}>"


def _launch(label: str, flops: float, fn):
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((label, flops, e0, e1))
    return r


def _dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return PS_F32
    if d == torch.bfloat16:
        return PS_BF16
    if d == torch.float16:
        return PS_F16
    raise TypeError(f"unsupported dtype {d}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_gpu(*ts: Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PsError("pistoseg_amd ops need device tensors: the HIP path has no CPU fallback")


def _ldc(t: Tensor) -> int:
    """Channel stride of a channels-last activation [N,H,W,C] (pixels must be densely packed rows of ldc)."""
    assert t.dim() == 4 and t.stride(3) == 1, "activation must be [N,H,W,C] with unit channel stride"
    ldc = t.stride(2)
    assert t.stride(1) == ldc * t.shape[2] and t.stride(0) == ldc * t.shape[2] * t.shape[1], "pixels must be contiguous"
    return ldc


@dataclass
class ConvSpec:
    """Geometry of one convolution of the net (forward sense)."""

    cin: int
    cout: int
    ksize: int
    stride: int = 1
    dilation: int = 1

    def out_hw(self, h: int, w: int):
        return (h - 1) // self.stride + 1, (w - 1) // self.stride + 1


# Launch options of the conv kernels.  They are arguments of every C-ABI launch (ps_conv_geom.tiles_per_block / .gpu_shared, and the
# choice between ps_conv2d_wgrad and ps_conv2d_wgrad_det), and on the host side they are STATE OF THE CALLER: every model owns a
# `LaunchOpts` (resnet38d.Net.launch) that its plans pass to each launch, its trainer sets `deterministic` on it and its gradient
# reducer `tiles_per_block` -- two trainers in one process do not see each other's switches.  A field left at None falls back to the
# module-level default below, which only direct callers of this module (the op tests, the probes under tools/) ever change.
@dataclass
class LaunchOpts:
    # 0 = one block per CU lives for the whole launch; n > 0 while a communication kernel shares the GPU (dist.BucketedAllReduce sets it on
    # its model's options between its first bucket launch and finish())
    tiles_per_block: Optional[int] = None
    # 1 while another stream of the caller runs kernels beside the launch stream (the backbone's two-stream backward, for its own duration): a
    # launch's partial last round is then left to the co-running kernel's blocks instead of being re-issued as smaller tiles
    gpu_shared: Optional[int] = None
    # weight gradients without atomics / top-k ties in index order (see DETERMINISTIC below); None = the module default = torch's switch
    deterministic: Optional[bool] = None
    # CUs left to a co-running kernel that holds them (an RCCL collective): the persistent kernels' grid and static schedule are sized for
    # the rest (ps_conv_geom.cus_reserved); set by dist.BucketedAllReduce(share="reserve") while buckets are in flight
    cus_reserved: Optional[int] = None
    # split precisions: plane pairs the weight gradient multiplies (ps_conv_geom.wgrad_terms): None / 1 = hi halves only, 3 = all three terms
    wgrad_terms: Optional[int] = None
    # 1 = the persistent kernels draw every work item behind a block's first one from per-XCD ticket counters (ps_conv_geom.tile_queue): blocks
    # that start late or share their CU take fewer items; set by dist.BucketedAllReduce(share="queue"), or for good on a model whose GPU is shared
    tile_queue: Optional[int] = None
    # stream-K finish of a launch's partial last round (ps_epilogue.sk_ws): None = the module default (on).  It re-associates the f32 sums at
    # its split points and only runs on the static schedule, so a model whose OTHER launch options change from launch to launch by timing (the
    # gradient reducer's tile_queue / cus_reserved while buckets are in flight) switches it off: results must not depend on when a bucket finished
    stream_k: Optional[bool] = None


TILES_PER_BLOCK = 0  # module defaults (see LaunchOpts): read when a launch is ENQUEUED
GPU_SHARED = 0
CUS_RESERVED = 0
TILE_QUEUE = 0


def _geom(spec: ConvSpec, dtype: int, n: int, h: int, w: int, ldc_x: int, ldc_y: int, opts: Optional[LaunchOpts] = None) -> ConvGeom:
    tpb = TILES_PER_BLOCK if opts is None or opts.tiles_per_block is None else opts.tiles_per_block
    shared = GPU_SHARED if opts is None or opts.gpu_shared is None else opts.gpu_shared
    reserved = CUS_RESERVED if opts is None or opts.cus_reserved is None else opts.cus_reserved
    terms = 0 if opts is None or opts.wgrad_terms is None else opts.wgrad_terms
    queue = TILE_QUEUE if opts is None or opts.tile_queue is None else opts.tile_queue
    return ConvGeom(dtype, n, h, w, spec.cin, spec.cout, spec.ksize, spec.stride, spec.dilation, ldc_x, ldc_y, int(tpb), int(shared), int(reserved), int(terms),
                    int(queue))


SPLIT_WIDTH = 2  # stored 16-bit channels per logical channel of a split tensor (hi + lo)


def _conv_dt(t: Tensor, split: bool) -> int:
    """C-ABI dtype of a conv launch on activation tensor t: `split` marks bf16 / fp16 tensors in the split layout (PS_BF16X3 / PS_F16X3: 2 C
    stored channels, blocks of 32 logical channels as [hi(32) | lo(32)])."""
    if split:
        assert t.dtype in (torch.bfloat16, torch.float16), "split tensors are bf16 or fp16 planes"
        return PS_BF16X3 if t.dtype == torch.bfloat16 else PS_F16X3
    return _dt(t)


def attach_hi(split_t: Tensor, hi: Tensor) -> Tensor:
    """Give a split tensor [..., 2C] a plain 16-bit companion [..., C] for its hi halves.  The conv epilogue that writes the split tensor as its
    `out` fills the companion as well (ps_epilogue.out_hi), and a split weight gradient whose operands BOTH carry one runs the plain 16-bit kernel
    on the companions: the same x_hi dy_hi term from contiguous rows instead of a gather of every other 64 bytes (1.6 x faster per launch).
    The companion hangs on the tensor OBJECT (slices are registered separately), so no address can be confused with a recycled one."""
    assert hi.dtype == split_t.dtype and tuple(hi.shape[:-1]) == tuple(split_t.shape[:-1]) and 2 * hi.shape[-1] == split_t.shape[-1]
    split_t._ps_hi = hi
    return split_t


def _hi_of(t: Optional[Tensor]) -> Optional[Tensor]:
    return getattr(t, "_ps_hi", None) if t is not None else None


# Scratch of the stream-K finish (ps_epilogue.sk_ws): one grow-only ZEROED buffer per (device, stream) -- the launches of a stream run in order and
# every launch leaves its arrival counters zeroed.  Sizes per geometry are asked of the library once and remembered.
_SK_WS: dict = {}
_SK_NEED: dict = {}
STREAM_K = os.environ.get("PISTOSEG_STREAM_K", "1") != "0"  # module switch (tools' A/Bs, PISTOSEG_STREAM_K=0): False = never hand a workspace to the conv launches


def _sk_workspace(g: ConvGeom, dgrad: bool, device, opts: Optional["LaunchOpts"] = None):
    if not (STREAM_K if opts is None or opts.stream_k is None else opts.stream_k):
        return None
    key = (g.dtype, g.n, g.h, g.w, g.cin, g.cout, g.ksize, g.stride, g.dilation, g.tiles_per_block, g.gpu_shared, g.cus_reserved, g.tile_queue, dgrad)
    need = _SK_NEED.get(key)
    if need is None:
        need = _SK_NEED[key] = int(_lib.load().ps_conv_sk_workspace_bytes(C.byref(g), 1 if dgrad else 0)) if g.ksize == 3 and g.stride == 1 else 0
    if need <= 0:
        return None
    wkey = (device, _stream())
    ws = _SK_WS.get(wkey)
    if ws is None or ws.numel() < need:
        ws = _SK_WS[wkey] = torch.zeros(max(need, 64 << 20), device=device, dtype=torch.uint8)  # (allocated on, and zeroed by, the launch stream)
    return ws


def _epilogue(mode=PS_EPI_NONE, add0=None, out_raw=None, scale=None, shift=None, drop=None, mask_src=None, add1=None, out=None, out_hi=None) -> Epilogue:
    e = Epilogue()
    e.mode = mode
    if add0 is not None:
        e.add0, e.ldc_add0 = add0.data_ptr(), _ldc(add0)
    if out_raw is not None:
        e.out_raw, e.ldc_raw = out_raw.data_ptr(), _ldc(out_raw)
    e.scale, e.shift, e.drop = _ptr(scale), _ptr(shift), _ptr(drop)
    if mask_src is not None:
        e.mask_src, e.ldc_mask = mask_src.data_ptr(), _ldc(mask_src)
    if add1 is not None:
        e.add1, e.ldc_add1 = add1.data_ptr(), _ldc(add1)
    if out is not None:
        e.out, e.ldc_out = out.data_ptr(), _ldc(out)
        if out_hi is not None:
            e.out_hi, e.ldc_hi = out_hi.data_ptr(), _ldc(out_hi)
    return e


def conv2d_fwd(spec: ConvSpec, x: Tensor, w_fwd: Tensor, *, add0=None, out_raw=None, bn_scale=None, bn_shift=None, drop=None,
               out_act=None, relu=False, split: bool = False, opts: Optional[LaunchOpts] = None) -> None:
    """y = conv(x, W) [+ add0]; out_raw <- y; out_act <- max(y*scale+shift, 0)*drop (if out_act given).
    split: every activation tensor is in the split layout (2 stored channels per logical one) and w_fwd is [cout][taps][2 cin] (PS_BF16X3 / PS_F16X3)."""
    _require_gpu(x, w_fwd)
    n, h, w, c = x.shape
    pl = SPLIT_WIDTH if split else 1
    assert c == pl * spec.cin and w_fwd.numel() == pl * spec.cout * spec.cin * spec.ksize**2 and w_fwd.dtype == x.dtype
    mode = PS_EPI_BNRELU if out_act is not None else PS_EPI_NONE
    ref = out_act if out_act is not None else out_raw
    g = _geom(spec, _conv_dt(x, split), n, h, w, _ldc(x), _ldc(ref), opts)
    e = _epilogue(mode, add0=add0, out_raw=out_raw, scale=bn_scale, shift=bn_shift, drop=drop, out=out_act, out_hi=_hi_of(out_act) if split else None)
    ws = _sk_workspace(g, False, x.device, opts)
    if ws is not None:
        e.sk_ws, e.sk_ws_bytes = ws.data_ptr(), ws.numel()
    lib = _lib.load()
    ho, wo = spec.out_hw(h, w)
    m = n * ho * wo
    _launch(_conv_label("fwd", g) if PROFILE is not None else "", 2.0 * m * spec.cout * spec.cin * spec.ksize**2,  # (algorithmic FLOPs: a split launch spends three MFMAs per product)
            lambda: _lib.check(lib.ps_conv2d_fwd(C.byref(g), x.data_ptr(), w_fwd.data_ptr(), C.byref(e), _stream()), "ps_conv2d_fwd"))


def conv2d_dgrad(spec: ConvSpec, dy: Tensor, w_dgrad: Tensor, x_hw, *, add0=None, out_raw=None, mask_src=None, bn_scale=None,
                 drop=None, add1=None, out=None, split: bool = False, opts: Optional[LaunchOpts] = None) -> None:
    """dx = conv^T(dy, W) [+ add0]; out_raw <- dx; out <- (mask_src>0 ? dx*scale*drop : 0) [+ add1].  split: as conv2d_fwd."""
    _require_gpu(dy, w_dgrad)
    n = dy.shape[0]
    h, w = x_hw
    assert dy.shape[3] == (SPLIT_WIDTH if split else 1) * spec.cout and tuple(dy.shape[1:3]) == spec.out_hw(h, w) and w_dgrad.dtype == dy.dtype
    mode = PS_EPI_RELUBWD if out is not None else PS_EPI_NONE
    ref = out if out is not None else out_raw
    g = _geom(spec, _conv_dt(dy, split), n, h, w, _ldc(ref), _ldc(dy), opts)
    e = _epilogue(mode, add0=add0, out_raw=out_raw, scale=bn_scale, drop=drop, mask_src=mask_src, add1=add1, out=out, out_hi=_hi_of(out) if split else None)
    ws = _sk_workspace(g, True, dy.device, opts)
    if ws is not None:
        e.sk_ws, e.sk_ws_bytes = ws.data_ptr(), ws.numel()
    lib = _lib.load()
    mo = n * dy.shape[1] * dy.shape[2]
    _launch(_conv_label("dgrad", g) if PROFILE is not None else "", 2.0 * mo * spec.cout * spec.cin * spec.ksize**2,
            lambda: _lib.check(lib.ps_conv2d_dgrad(C.byref(g), dy.data_ptr(), w_dgrad.data_ptr(), C.byref(e), _stream()), "ps_conv2d_dgrad"))


def conv1x1_head_supported(spec: ConvSpec, x: Tensor, classes: int) -> int:
    """Workspace floats ps_conv1x1_head_fwd needs for this launch, 0 if the fused kernel does not serve it (f32, narrow layers, ...)."""
    if spec.ksize != 1 or spec.stride != 1 or x.dtype == torch.float32:
        return 0
    n, h, w, _ = x.shape
    g = _geom(spec, _dt(x), n, h, w, _ldc(x), spec.cout)
    return int(_lib.load().ps_conv1x1_head_workspace_floats(C.byref(g), classes))


_HEAD_WS: dict = {}


def conv1x1_head_fwd(spec: ConvSpec, x: Tensor, w_fwd: Tensor, bn_scale: Tensor, bn_shift: Tensor, w_head: Tensor, cam: Tensor) -> None:
    """cam[n,h,w,C] (f32) = head(T(relu(bn(conv1x1(x))))) in one launch + an ordered reduction: the activated tensor is never written
    (inference: b7's last conv + bn7 + ReLU + fc8, resnet38d.py:186 / revise_net.py:50).  w_head: [C, cout] f32."""
    _require_gpu(x, w_fwd, w_head, cam)
    n, h, w, _ = x.shape
    classes = w_head.shape[0]
    assert w_head.shape == (classes, spec.cout) and w_head.dtype == torch.float32 and w_head.is_contiguous()
    assert cam.shape == (n, h, w, classes) and cam.dtype == torch.float32 and cam.is_contiguous()
    g = _geom(spec, _dt(x), n, h, w, _ldc(x), spec.cout)
    lib = _lib.load()
    need = int(lib.ps_conv1x1_head_workspace_floats(C.byref(g), classes))
    if need <= 0:
        raise _lib.PsError("conv1x1_head_fwd: geometry not served by the fused kernel")
    key = (x.device, _stream())  # grow-only, one per (device, stream): launches of one stream run in order and may share it
    ws = _HEAD_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _HEAD_WS[key] = torch.empty(need, device=x.device, dtype=torch.float32)
    _launch(f"conv_gemm256_kernel<{ {PS_BF16: 'bf16', PS_F16: 'f16'}.get(g.dtype, 'f32') }>" if PROFILE is not None else "",
            2.0 * n * h * w * spec.cout * spec.cin,
            lambda: _lib.check(lib.ps_conv1x1_head_fwd(C.byref(g), x.data_ptr(), w_fwd.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(),
                                                       w_head.data_ptr(), classes, ws.data_ptr(), need, cam.data_ptr(), _stream()),
                               "ps_conv1x1_head_fwd"))


# Deterministic weight gradients (ps_conv2d_wgrad_det): partial sums of the pixel ranges go to a workspace and are added up in range
# order instead of by f32 atomics -- the reference's `torch.use_deterministic_algorithms(True)` / `Trainer(deterministic=True)`
# (revise_pseudo_labels.py:140-146, segmentation_train.py:153-160).  None = follow torch's own switch, i.e. exactly what those two
# reference call sites set (`torch.are_deterministic_algorithms_enabled()`); True / False force it (the native trainers'
# `deterministic=` argument).  Read when a launch is enqueued.  The workspace is one grow-only buffer per (device, stream): launches
# on one stream run in order and may share it.
DETERMINISTIC: Optional[bool] = None
_WGRAD_WS: dict = {}


def deterministic_enabled(override: Optional[bool] = None) -> bool:
    if override is not None:
        return override
    return torch.are_deterministic_algorithms_enabled() if DETERMINISTIC is None else DETERMINISTIC


def _wgrad_workspace(nbytes: int, device, stream: int) -> Tensor:
    """`stream`: the raw handle the launch is enqueued on (the same lookup as the launch itself: ops._stream())."""
    key = (device, stream)
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)  # allocated on (= owned by) the current stream
        _WGRAD_WS[key] = ws
    return ws


def release_workspaces() -> None:
    """Drop the grow-only scratch buffers of this module (deterministic weight-gradient slices, fc8 partial sums, the fused head's
    partial sums): trainers call it on teardown; the next launch that needs one allocates it again."""
    _WGRAD_WS.clear()
    _fc8_ws.clear()
    _HEAD_WS.clear()
    _SK_WS.clear()
    if torch.cuda.is_available():
        torch.cuda.synchronize()  # (no queue-mode launch in flight)
        _lib.check(_lib.load().ps_queue_release(), "ps_queue_release")


def conv2d_wgrad(spec: ConvSpec, x: Tensor, dy: Tensor, dw: Tensor, deterministic: Optional[bool] = None, split: bool = False,
                 opts: Optional[LaunchOpts] = None) -> None:
    """dw[cout][kh][kw][cin] (f32, channels-last OIHW storage) += sum_pixels dy * x@tap.  deterministic (default: opts.deterministic, then the
    module switch DETERMINISTIC): no atomics, bit-identical from run to run.
    split: x / dy are split tensors: dw += x_hi dy_hi [+ x_hi dy_lo + x_lo dy_hi with opts.wgrad_terms = 3], launches of the 16-bit kernels on the
    hi / lo halves inside the one C-ABI call (the weight gradient contracts over PIXELS, so hi and lo cannot share a K-line as they do in the
    forward / data-gradient kernels; the lo terms are below the gradient's own noise: include/pistoseg_hip.h, ps_conv_geom.wgrad_terms)."""
    _require_gpu(x, dy, dw)
    if split and _hi_of(x) is not None and _hi_of(dy) is not None and (opts is None or opts.wgrad_terms in (None, 0, 1)):
        x, dy, split = _hi_of(x), _hi_of(dy), False  # the hi halves as plain 16-bit tensors (attach_hi): the x_hi dy_hi term on contiguous rows
    n, h, w, c = x.shape
    pl = SPLIT_WIDTH if split else 1
    assert c == pl * spec.cin and dy.shape[3] == pl * spec.cout and dw.dtype == torch.float32 and x.dtype == dy.dtype
    assert dw.numel() == spec.cout * spec.cin * spec.ksize**2
    g = _geom(spec, _conv_dt(x, split), n, h, w, _ldc(x), _ldc(dy), opts)
    lib = _lib.load()
    mo = n * dy.shape[1] * dy.shape[2]
    flops = 2.0 * mo * spec.cout * spec.cin * spec.ksize**2  # (algorithmic)
    if deterministic is None and opts is not None:
        deterministic = opts.deterministic
    if deterministic_enabled(deterministic):
        need = int(lib.ps_conv2d_wgrad_det_workspace_bytes(C.byref(g)))
        if need < 0:
            _lib.check(-1, "ps_conv2d_wgrad_det_workspace_bytes")
        stream = _stream()
        ws = _wgrad_workspace(need, x.device, stream) if need > 0 else None
        _launch(_conv_label("wgrad", g) if PROFILE is not None else "", flops,
                lambda: _lib.check(lib.ps_conv2d_wgrad_det(C.byref(g), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(ws), need, stream),
                                   "ps_conv2d_wgrad_det"))
        return
    _launch(_conv_label("wgrad", g) if PROFILE is not None else "", flops,
            lambda: _lib.check(lib.ps_conv2d_wgrad(C.byref(g), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _stream()), "ps_conv2d_wgrad"))


def weight_transpose(src: Tensor, dst: Tensor, cout: int, taps: int, cin: int) -> None:
    _require_gpu(src, dst)
    assert src.numel() == dst.numel() == cout * taps * cin
    lib = _lib.load()
    _lib.check(lib.ps_weight_transpose(_dt(src), _dt(dst), src.data_ptr(), dst.data_ptr(), cout, taps, cin, _stream()), "ps_weight_transpose")


def weight_transpose_batched(items) -> None:
    """items: (src [cout][taps][cin], dst view with rows [cin*taps] of stride dst.stride(0) >= cout, cout, taps, cin), all of one
    dtype pair.  One launch for the whole list (ps_weight_transpose_batched)."""
    if not items:
        return
    arr = (_lib.WtItem * len(items))()
    sdt, ddt = _dt(items[0][0]), _dt(items[0][1])
    for k, (src, dst, cout, taps, cin) in enumerate(items):
        _require_gpu(src, dst)
        assert _dt(src) == sdt and _dt(dst) == ddt and src.is_contiguous() and src.numel() == cout * taps * cin
        assert dst.dim() == 2 and dst.shape == (cin * taps, cout) and dst.stride(1) == 1
        arr[k] = _lib.WtItem(src.data_ptr(), dst.data_ptr(), cout, taps, cin, dst.stride(0))
    lib = _lib.load()
    _lib.check(lib.ps_weight_transpose_batched(sdt, ddt, len(items), arr, _stream()), "ps_weight_transpose_batched")


def copy_rows(src: Tensor, dst: Tensor) -> None:
    """dst[r, :] = src[r, :] for 2-d views with unit inner stride and arbitrary row pitches (same dtype, same shape)."""
    _require_gpu(src, dst)
    assert src.dim() == 2 and src.shape == dst.shape and src.dtype == dst.dtype and src.stride(1) == 1 and dst.stride(1) == 1
    es = src.element_size()
    lib = _lib.load()
    _lib.check(lib.ps_copy_rows(src.data_ptr(), src.stride(0) * es, dst.data_ptr(), dst.stride(0) * es, src.shape[0], src.shape[1] * es, _stream()),
               "ps_copy_rows")


def convert_rows(src: Tensor, dst: Tensor, c: int, *, src_split: bool = False, dst_split: bool = False, weights: bool = False) -> None:
    """Row-wise conversion between f32 and a storage format (ps_convert_rows): src / dst are [..., C'] views with unit channel stride whose
    leading dims are densely packed rows of pitch stride(-2); c = logical channels.  f32 -> bf16 / fp16 / split layout and back.
    (weights: kept for call-site readability; activations and weights share the split layout.)"""
    _require_gpu(src, dst)
    assert src.stride(-1) == 1 and dst.stride(-1) == 1
    rows = src.numel() // src.shape[-1]
    assert rows == dst.numel() // dst.shape[-1]

    def pitch(t):
        if t.dim() == 1:
            return t.shape[0]
        ld = t.stride(-2)
        exp = ld
        for d in range(t.dim() - 2, -1, -1):  # rows must be equally spaced
            assert t.stride(d) == exp or t.shape[d] == 1, "rows must be densely packed"
            exp *= t.shape[d]
        return ld

    sf = _conv_dt(src, True) if src_split else _dt(src)
    df = _conv_dt(dst, True) if dst_split else _dt(dst)
    assert src.shape[-1] == (SPLIT_WIDTH if src_split else 1) * c and dst.shape[-1] == (SPLIT_WIDTH if dst_split else 1) * c
    lib = _lib.load()
    _lib.check(lib.ps_convert_rows(src.data_ptr(), sf, pitch(src), dst.data_ptr(), df, pitch(dst), rows, c, int(weights), _stream()), "ps_convert_rows")


def cast_f32_bf16(src: Tensor, dst: Tensor) -> None:
    _require_gpu(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    lib = _lib.load()
    _lib.check(lib.ps_cast_f32_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "ps_cast_f32_bf16")


def cast_f32_lowp(src: Tensor, dst: Tensor) -> None:
    """dst (bf16 or fp16) = cast(src f32): refresh of the 16-bit forward weights from the f32 master arena."""
    _require_gpu(src, dst)
    assert src.dtype == torch.float32 and dst.dtype in (torch.bfloat16, torch.float16) and src.numel() == dst.numel()
    lib = _lib.load()
    _lib.check(lib.ps_cast_f32_lowp(src.data_ptr(), dst.data_ptr(), _dt(dst), src.numel(), _stream()), "ps_cast_f32_lowp")


def conv_front_s2(x_nchw: Tensor, w1a: Tensor, scale0: Tensor, shift0: Tensor, w_b1: Tensor, w_2a: Tensor, out_b1: Tensor, scale1: Tensor,
                  shift1: Tensor, out_2a: Tensor) -> None:
    """EXPERIMENT (debug library only: `_lib.use_debug_library()` first; correct, not faster than the launches it replaces).  conv1a + BN + ReLU
    and the first ResBlock's two stride-2 convs (1x1 shortcut, raw; 3x3 + BN + ReLU) in one launch: conv1a's activation is never written
    (ps_debug_conv_front_s2; resnet38d.py:123,161-162 + ResBlock.forward :28-41).  w_b1 [128][1][1][64], w_2a [128][3][3][64] in the outputs'
    16-bit type; outputs channels-last [n, h/2, w/2, 128] (channel slices of wider buffers are fine)."""
    _require_gpu(x_nchw, w1a, w_b1, w_2a, out_b1, out_2a)
    n, _, h, w = x_nchw.shape
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous() and w1a.dtype == torch.float32 and w1a.numel() == 64 * 27
    assert w_b1.dtype == w_2a.dtype == out_b1.dtype == out_2a.dtype and w_b1.numel() == 128 * 64 and w_2a.numel() == 128 * 9 * 64
    assert out_b1.shape == out_2a.shape == (n, h // 2, w // 2, 128)
    lib = _lib.load()
    assert hasattr(lib, "ps_debug_conv_front_s2"), "conv_front_s2 lives in the debug library (_lib.use_debug_library())"
    assert lib.ps_debug_conv_front_s2_supported(_dt(out_2a), n, h, w)
    _lib.check(lib.ps_debug_conv_front_s2(_dt(out_2a), n, h, w, x_nchw.data_ptr(), w1a.data_ptr(), scale0.data_ptr(), shift0.data_ptr(), w_b1.data_ptr(),
                                          w_2a.data_ptr(), out_b1.data_ptr(), _ldc(out_b1), scale1.data_ptr(), shift1.data_ptr(), out_2a.data_ptr(),
                                          _ldc(out_2a), _stream()), "ps_debug_conv_front_s2")


def conv1a_fwd(x_nchw: Tensor, w_oihw: Tensor, bn_scale: Optional[Tensor], bn_shift: Optional[Tensor], out_act: Optional[Tensor],
               out_raw: Optional[Tensor] = None) -> None:
    _require_gpu(x_nchw, w_oihw)
    n, c, h, w = x_nchw.shape
    assert c == 3 and x_nchw.is_contiguous() and x_nchw.dtype == torch.float32
    assert w_oihw.is_contiguous() and w_oihw.dtype == torch.float32 and tuple(w_oihw.shape) == (64, 3, 3, 3)
    ref = out_act if out_act is not None else out_raw
    assert ref.is_contiguous() and tuple(ref.shape) == (n, h, w, 64)
    lib = _lib.load()
    _lib.check(
        lib.ps_conv1a_fwd(_dt(ref), x_nchw.data_ptr(), w_oihw.data_ptr(), _ptr(bn_scale), _ptr(bn_shift), _ptr(out_act), _ptr(out_raw), n, h, w, _stream()),
        "ps_conv1a_fwd",
    )


def fc8_fwd(x: Tensor, w: Tensor, drop: Optional[Tensor], cam: Tensor) -> None:
    """cam[N,g,g,C] (f32) = fc8(dropout7(x)); x channels-last [N,g,g,K]; w f32 [C,K]."""
    _require_gpu(x, w, cam)
    n, h, wd, k = x.shape
    c = w.shape[0]
    assert w.dtype == torch.float32 and w.is_contiguous() and cam.dtype == torch.float32 and cam.is_contiguous()
    assert tuple(cam.shape) == (n, h, wd, c)
    lib = _lib.load()
    _lib.check(lib.ps_fc8_fwd(_dt(x), x.data_ptr(), _ldc(x), w.data_ptr(), _ptr(drop), cam.data_ptr(), n * h * wd, h * wd, k, c, _stream()), "ps_fc8_fwd")


def fc_head_fwd(x: Tensor, w: Tensor, ldw: int, bias: Optional[Tensor], drop: Optional[Tensor], cam: Tensor, accumulate: bool) -> None:
    """cam[N,g,g,C] (f32) (=|+=) x @ w[:, :K]^T + bias; w: f32 view whose rows are ldw floats apart (a column slice of a wider head)."""
    _require_gpu(x, w, cam)
    n, h, wd, k = x.shape
    c = cam.shape[3]
    assert w.dtype == torch.float32 and cam.dtype == torch.float32 and cam.is_contiguous() and tuple(cam.shape) == (n, h, wd, c)
    lib = _lib.load()
    _lib.check(lib.ps_fc_head_fwd(_dt(x), x.data_ptr(), _ldc(x), w.data_ptr(), int(ldw), _ptr(bias), _ptr(drop), cam.data_ptr(), int(accumulate),
                                  n * h * wd, h * wd, k, c, _stream()), "ps_fc_head_fwd")


_fc8_ws = {}  # device -> workspace tensor for the dw partial sums (grown on demand, reused across steps)


def fc8_bwd(x: Tensor, w: Tensor, drop: Optional[Tensor], scale7: Tensor, dcam: Tensor, dx: Tensor, dw: Tensor) -> None:
    _require_gpu(x, w, dcam, dx, dw)
    n, h, wd, k = x.shape
    c = w.shape[0]
    assert dcam.is_contiguous() and dcam.dtype == torch.float32 and dw.dtype == torch.float32 and dw.is_contiguous()
    lib = _lib.load()
    need = int(lib.ps_fc8_bwd_workspace_floats(n * h * wd, h * wd, k, c))
    ws = _fc8_ws.get(x.device)
    if ws is None or ws.numel() < need:
        ws = _fc8_ws[x.device] = torch.empty(need, device=x.device, dtype=torch.float32)
    _lib.check(
        lib.ps_fc8_bwd_ws(_dt(x), x.data_ptr(), _ldc(x), w.data_ptr(), _ptr(drop), scale7.data_ptr(), dcam.data_ptr(), dx.data_ptr(), _ldc(dx),
                          dw.data_ptr(), n * h * wd, h * wd, k, c, ws.data_ptr(), ws.numel(), _stream()),
        "ps_fc8_bwd_ws",
    )


def _t4(t: Tensor, layout: str) -> Tensor4:
    """View a 4-d tensor as (n, c, h, w) with element strides. layout: 'nchw' or 'nhwc' (how t's dims are ordered)."""
    if layout == "nchw":
        n, c, h, w = t.shape
        sn, sc, sh, sw = t.stride()
    else:
        n, h, w, c = t.shape
        sn, sh, sw, sc = t.stride()
    return Tensor4(t.data_ptr(), _dt(t), n, c, h, w, 0, sn, sc, sh, sw)


def bilinear_fwd(src: Tensor, src_layout: str, dst: Tensor, dst_layout: str, align_corners: bool) -> None:
    _require_gpu(src, dst)
    a, b = _t4(src, src_layout), _t4(dst, dst_layout)
    lib = _lib.load()
    _lib.check(lib.ps_bilinear_fwd(C.byref(a), C.byref(b), int(align_corners), _stream()), "ps_bilinear_fwd")


def bilinear_bwd(ddst: Tensor, ddst_layout: str, dsrc: Tensor, dsrc_layout: str, align_corners: bool) -> None:
    _require_gpu(ddst, dsrc)
    a, b = _t4(ddst, ddst_layout), _t4(dsrc, dsrc_layout)
    lib = _lib.load()
    _lib.check(lib.ps_bilinear_bwd(C.byref(a), C.byref(b), int(align_corners), _stream()), "ps_bilinear_bwd")


def softmax_ce(logits: Tensor, target: Tensor, ignore_index: Optional[int], want_grad: bool, grad_scale: float = 1.0):
    """Mean-over-all-pixels CE (SegmentationModule.training_step).  Returns (loss[1] f32, dlogits or None)."""
    _require_gpu(logits, target)
    n, c, h, w = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32 and target.dtype == torch.int64 and target.is_contiguous()
    lib = _lib.load()
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    partials = torch.empty(int(lib.ps_ce_workspace_floats()), device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(
        lib.ps_softmax_ce(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), _ptr(dlogits), float(grad_scale), n, c, h, w,
                          -1 if ignore_index is None else int(ignore_index), partials.data_ptr(), _stream()),
        "ps_softmax_ce",
    )
    return loss, dlogits


def dice_loss(logits: Tensor, target: Tensor, ignore_index: Optional[int], want_grad: bool, grad_scale: float = 1.0):
    """smp-style multiclass Dice (MosaicModule.training_step).  Returns (loss[1] f32, dlogits or None)."""
    _require_gpu(logits, target)
    n, c, h, w = logits.shape
    assert logits.is_contiguous() and logits.dtype == torch.float32 and target.dtype == torch.int64 and target.is_contiguous()
    lib = _lib.load()
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    ws = torch.empty(int(lib.ps_dice_workspace_floats()), device=logits.device, dtype=torch.float32)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(
        lib.ps_dice_loss(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), _ptr(dlogits), float(grad_scale), n, c, h, w,
                         -1 if ignore_index is None else int(ignore_index), ws.data_ptr(), _stream()),
        "ps_dice_loss",
    )
    return loss, dlogits


def argmax_mask(x: Tensor, *, mode: int = _lib.PS_MASK_PLAIN, softmax_first: bool = False, first_ch: int = 0, label: Optional[Tensor] = None,
                tissue: Optional[Tensor] = None, want_entropy: bool = False):
    """NCHW f32 scores -> uint8 mask [N,H,W] (+ optional f32 entropy)."""
    _require_gpu(x)
    n, c, h, w = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    mask = torch.empty((n, h, w), device=x.device, dtype=torch.uint8)
    ent = torch.empty((n, h, w), device=x.device, dtype=torch.float32) if want_entropy else None
    if label is not None:
        label = label.reshape(n, c).to(torch.float32).contiguous()
    if tissue is not None:
        assert tissue.dtype == torch.uint8 and tissue.is_contiguous() and tuple(tissue.shape) == (n, h, w)
    lib = _lib.load()
    _lib.check(
        lib.ps_argmax_mask(x.data_ptr(), _ptr(label), _ptr(tissue), mask.data_ptr(), _ptr(ent), mode, int(softmax_first), first_ch, n, c, h, w, _stream()),
        "ps_argmax_mask",
    )
    return (mask, ent) if want_entropy else mask


def confusion_accum(pred: Tensor, gt: Tensor, cm: Tensor, num_class: int) -> None:
    _require_gpu(pred, gt, cm)
    assert pred.dtype == torch.uint8 and gt.dtype == torch.int64 and cm.dtype == torch.int64 and cm.numel() == num_class * num_class
    assert pred.is_contiguous() and gt.is_contiguous() and pred.numel() == gt.numel()
    lib = _lib.load()
    _lib.check(lib.ps_confusion_accum(pred.data_ptr(), gt.data_ptr(), cm.data_ptr(), pred.numel(), num_class, _stream()), "ps_confusion_accum")


def iou_from_confusion(cm: Tensor, num_class: int, out: Optional[Tensor] = None) -> Tensor:
    """f64 [2 + num_class] on the device: (mIoU, fwIoU, per-tissue IoU...) of the int64 confusion matrix, loss.py:28-53 in numpy's evaluation order."""
    _require_gpu(cm)
    assert cm.dtype == torch.int64 and cm.numel() == num_class * num_class and cm.is_contiguous()
    if out is None:
        out = torch.empty(2 + num_class, device=cm.device, dtype=torch.float64)
    assert out.dtype == torch.float64 and out.numel() == 2 + num_class and out.is_contiguous()
    lib = _lib.load()
    _lib.check(lib.ps_iou_from_confusion(cm.data_ptr(), num_class, out.data_ptr(), _stream()), "ps_iou_from_confusion")
    return out


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, p_shadow: Optional[Tensor], lr: float, betas, eps: float, weight_decay: float, step: int,
               grad_inv_scale: float = 1.0) -> None:
    """Fused AdamW over a flat arena; p_shadow (bf16 or fp16, optional) receives the refreshed 16-bit weights;
    grad_inv_scale undoes an fp16 loss scale."""
    _require_gpu(p, g, m, v)
    lib = _lib.load()
    sdt = PS_BF16 if p_shadow is None else _dt(p_shadow)
    _lib.check(
        lib.ps_adamw_step_scaled(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_shadow), sdt, p.numel(), lr, betas[0], betas[1], eps,
                                 weight_decay, step, float(grad_inv_scale), _stream()),
        "ps_adamw_step_scaled",
    )


def adamw_step_guarded(p: Tensor, g: Tensor, m: Tensor, v: Tensor, p_shadow: Optional[Tensor], lr: float, betas, eps: float, weight_decay: float,
                       state: Tensor, grad_inv_scale: float = 1.0) -> None:
    """AdamW with the dynamic-loss-scale overflow check on the device (ps_adamw_step_guarded): state = int32[2] = (steps applied so far, non-finite
    elements of this step's gradient); a step whose gradient overflowed changes nothing, an applied one advances state[0]."""
    _require_gpu(p, g, m, v, state)
    assert state.dtype == torch.int32 and state.numel() == 2 and state.is_contiguous()
    lib = _lib.load()
    sdt = PS_BF16 if p_shadow is None else _dt(p_shadow)
    _lib.check(
        lib.ps_adamw_step_guarded(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(p_shadow), sdt, p.numel(), lr, betas[0], betas[1], eps,
                                  weight_decay, state.data_ptr(), float(grad_inv_scale), _stream()),
        "ps_adamw_step_guarded",
    )


def dropout2d_masks(segments, n: int, device, seed: int, offset: int):
    """segments: [(name, channels, p)] -> {name: [n, channels] f32 multipliers (0 or 1/(1-p))}, all drawn by ONE launch into one flat
    buffer (Philox4x32-10 keyed by (seed, offset); nn.Dropout2d's Bernoulli draw, resnet38d.py:63,67,85,90, revise_net.py:11,50)."""
    assert 1 <= len(segments) <= 8
    plan = _lib.DropoutPlan()
    plan.nseg = len(segments)
    off = 0
    for k, (_, c, p) in enumerate(segments):
        off += n * c
        plan.end[k], plan.p[k] = off, p
    flat = torch.empty(off, device=device, dtype=torch.float32)
    _require_gpu(flat)
    lib = _lib.load()
    _lib.check(lib.ps_dropout2d_masks(flat.data_ptr(), C.byref(plan), seed & (2**64 - 1), offset & (2**64 - 1), _stream()), "ps_dropout2d_masks")
    out, lo = {}, 0
    for name, c, _ in segments:
        out[name] = flat[lo:lo + n * c].view(n, c)
        lo += n * c
    return out


def nonfinite_count(g: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """1-element int32 device tensor: number of inf/nan entries of the f32 tensor g (out: a zeroed 1-element int32 view to count into)."""
    _require_gpu(g)
    assert g.dtype == torch.float32 and g.is_contiguous()
    if out is None:
        out = torch.zeros(1, device=g.device, dtype=torch.int32)
    assert out.dtype == torch.int32 and out.numel() == 1
    lib = _lib.load()
    _lib.check(lib.ps_nonfinite_count(g.data_ptr(), g.numel(), out.data_ptr(), _stream()), "ps_nonfinite_count")
    return out


def sgd_step(p: Tensor, g: Tensor, buf: Optional[Tensor], p_shadow: Optional[Tensor], lr: float, momentum: float, weight_decay: float, first_step: bool,
             grad_inv_scale: float = 1.0) -> None:
    _require_gpu(p, g)
    lib = _lib.load()
    sdt = PS_BF16 if p_shadow is None else _dt(p_shadow)
    _lib.check(
        lib.ps_sgd_step_scaled(p.data_ptr(), g.data_ptr(), _ptr(buf), _ptr(p_shadow), sdt, p.numel(), lr, momentum, weight_decay, int(first_step),
                               float(grad_inv_scale), _stream()),
        "ps_sgd_step_scaled",
    )


def sgd_step_guarded(p: Tensor, g: Tensor, buf: Optional[Tensor], p_shadow: Optional[Tensor], lr: float, momentum: float, weight_decay: float,
                     state: Tensor, advance: bool, poly_max_step: int = 0, poly_power: float = 0.0, grad_inv_scale: float = 1.0) -> None:
    """SGD with the dynamic-loss-scale overflow check (and the poly LR schedule) on the device (ps_sgd_step_guarded): state = int32[2] =
    (steps applied so far, non-finite elements of this step's gradient); `advance` on the last parameter group of a step."""
    _require_gpu(p, g, state)
    assert state.dtype == torch.int32 and state.numel() == 2 and state.is_contiguous()
    lib = _lib.load()
    sdt = PS_BF16 if p_shadow is None else _dt(p_shadow)
    _lib.check(
        lib.ps_sgd_step_guarded(p.data_ptr(), g.data_ptr(), _ptr(buf), _ptr(p_shadow), sdt, p.numel(), lr, momentum, weight_decay, state.data_ptr(),
                                int(advance), int(poly_max_step), float(poly_power), float(grad_inv_scale), _stream()),
        "ps_sgd_step_guarded",
    )


# ---------------------------------------------------------------------------------------------------
# RFM head + feature-consistency losses (csrc/rfm_ops.hip)
# ---------------------------------------------------------------------------------------------------
def bgemm(A: Tensor, B: Tensor, Cm: Tensor, batch: int, M: int, N: int, K: int, sa, sb, sc, alpha: float = 1.0) -> None:
    """C[b][m][n] = alpha * sum_k A[b][m*sam + k*sak] * B[b][k*sbk + n*sbn]; sa=(sab,sam,sak), sb=(sbb,sbk,sbn), sc=(scb,scm,scn)."""
    _require_gpu(A, B, Cm)
    lib = _lib.load()
    _lib.check(lib.ps_bgemm(_dt(A), _dt(B), _dt(Cm), A.data_ptr(), B.data_ptr(), Cm.data_ptr(), batch, M, N, K, *sa, *sb, *sc, alpha, _stream()), "ps_bgemm")


def softmax_rows_(x: Tensor, rows: int, length: int) -> None:
    _require_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    lib = _lib.load()
    _lib.check(lib.ps_softmax_rows(x.data_ptr(), rows, length, _stream()), "ps_softmax_rows")


def rfm_apply(P: Tensor, V: Tensor, R: Tensor) -> None:
    """R[n,j,:] = sum_i P[n,j,i] * V[n,i,:]."""
    _require_gpu(P, V, R)
    n, np_, _ = P.shape
    cc = V.shape[2]
    assert P.is_contiguous() and V.is_contiguous() and R.is_contiguous() and P.dtype == V.dtype == R.dtype == torch.float32
    lib = _lib.load()
    _lib.check(lib.ps_rfm_apply(P.data_ptr(), V.data_ptr(), R.data_ptr(), n, np_, cc, _stream()), "ps_rfm_apply")


def affinity_softmax_bwd_(P: Tensor, dR: Tensor, V: Tensor, R: Tensor) -> None:
    _require_gpu(P, dR, V, R)
    n, np_, _ = P.shape
    assert all(t.is_contiguous() and t.dtype == torch.float32 for t in (P, dR, V, R))
    lib = _lib.load()
    _lib.check(lib.ps_affinity_softmax_bwd(P.data_ptr(), dR.data_ptr(), V.data_ptr(), R.data_ptr(), n, np_, V.shape[2], _stream()), "ps_affinity_softmax_bwd")


def norm_cam(src: Tensor, src_layout: str, dst: Tensor, dst_strides, mode: int, label: Optional[Tensor] = None) -> None:
    """dst (f32) <- normalised CAM; dst_strides = element strides over (sample, channel, pixel)."""
    _require_gpu(src, dst)
    a = _t4(src, src_layout)
    assert dst.dtype == torch.float32
    lib = _lib.load()
    _lib.check(lib.ps_norm_cam(C.byref(a), dst.data_ptr(), *dst_strides, _ptr(label), mode, _stream()), "ps_norm_cam")


def _loss_ws(dev) -> Tensor:
    return torch.empty(int(_lib.load().ps_loss_workspace_floats()), device=dev, dtype=torch.float32)


def l1_masked(a: Tensor, b: Tensor, label: Tensor, loss: Tensor, accumulate: bool, da=None, db=None, grad_scale: float = 1.0) -> None:
    _require_gpu(a, b, label, loss)
    n, c, h, w = a.shape
    lib = _lib.load()
    _lib.check(lib.ps_l1_masked(a.data_ptr(), b.data_ptr(), label.data_ptr(), _ptr(da), _ptr(db), loss.data_ptr(), int(accumulate), grad_scale,
                                n, c, h, w, _loss_ws(a.device).data_ptr(), _stream()), "ps_l1_masked")


def ecr_tensor(ref: Tensor, rv: Tensor, label: Tensor, out: Tensor) -> None:
    _require_gpu(ref, rv, label, out)
    n, c, h, w = rv.shape
    lib = _lib.load()
    _lib.check(lib.ps_ecr_tensor(ref.data_ptr(), rv.data_ptr(), label.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "ps_ecr_tensor")


def ecr_bwd(ref, rv, label, t, thr, take, drv, grad_scale: float, deterministic: Optional[bool] = None) -> None:
    n, c, h, w = rv.shape
    lib = _lib.load()
    if deterministic_enabled(deterministic):  # ties at the threshold taken in index order (ps_ecr_bwd_det) instead of first come, first served
        need = int(lib.ps_tie_workspace_ints(n, h, w))
        counts = torch.empty(need, device=rv.device, dtype=torch.int32)
        _lib.check(lib.ps_ecr_bwd_det(ref.data_ptr(), rv.data_ptr(), label.data_ptr(), t.data_ptr(), thr.data_ptr(), take.data_ptr(), counts.data_ptr(),
                                      need, drv.data_ptr(), grad_scale, n, c, h, w, _stream()), "ps_ecr_bwd_det")
        return
    counter = torch.zeros(n, device=rv.device, dtype=torch.int32)
    _lib.check(lib.ps_ecr_bwd(ref.data_ptr(), rv.data_ptr(), label.data_ptr(), t.data_ptr(), thr.data_ptr(), take.data_ptr(), counter.data_ptr(),
                              drv.data_ptr(), grad_scale, n, c, h, w, _stream()), "ps_ecr_bwd")


def topk_select(x: Tensor, k: int, largest: bool, relu: bool = False):
    """x: [rows, len] f32 contiguous -> (thr[rows] f32, take[rows] i32, sums[rows] f32)."""
    _require_gpu(x)
    rows, length = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    thr = torch.empty(rows, device=x.device, dtype=torch.float32)
    take = torch.empty(rows, device=x.device, dtype=torch.int32)
    sums = torch.empty(rows, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    need = int(lib.ps_topk_select_workspace_bytes(rows))
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)  # (caching allocator: no real allocation after the first step)
    _lib.check(lib.ps_topk_select_ws(x.data_ptr(), rows, length, k, int(largest), int(relu), thr.data_ptr(), take.data_ptr(), sums.data_ptr(),
                                     ws.data_ptr(), need, _stream()), "ps_topk_select_ws")
    return thr, take, sums


def sum_scaled(x: Tensor, scale: float, out: Tensor, accumulate: bool) -> None:
    lib = _lib.load()
    _lib.check(lib.ps_sum_scaled(x.data_ptr(), x.numel(), scale, out.data_ptr(), int(accumulate), _stream()), "ps_sum_scaled")


def gap(x: Tensor) -> Tensor:
    n, c, h, w = x.shape
    out = torch.empty((n, c), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.ps_gap(x.data_ptr(), out.data_ptr(), n * c, h * w, _stream()), "ps_gap")
    return out


def softmargin(gap_: Tensor, label: Tensor, loss: Tensor, accumulate: bool, want_grad: bool, grad_scale: float = 1.0):
    n, c = gap_.shape
    dgap = torch.empty_like(gap_) if want_grad else None
    lib = _lib.load()
    _lib.check(lib.ps_softmargin(gap_.data_ptr(), label.data_ptr(), _ptr(dgap), loss.data_ptr(), int(accumulate), grad_scale, n, c, _stream()), "ps_softmargin")
    return dgap


def gap_bwd(dgap: Tensor, dx: Tensor) -> None:
    n, c, h, w = dx.shape
    lib = _lib.load()
    _lib.check(lib.ps_gap_bwd(dgap.data_ptr(), dx.data_ptr(), n * c, h * w, _stream()), "ps_gap_bwd")


def chmax(x: Tensor, label: Tensor):
    n, c, h, w = x.shape
    m = torch.empty((n, h * w), device=x.device, dtype=torch.float32)
    arg = torch.empty((n, h * w), device=x.device, dtype=torch.uint8)
    lib = _lib.load()
    _lib.check(lib.ps_chmax(x.data_ptr(), label.data_ptr(), m.data_ptr(), arg.data_ptr(), n, c, h, w, _stream()), "ps_chmax")
    return m, arg


def minpool_bwd(m, arg, label, thr, take, dx, grad_scale: float, deterministic: Optional[bool] = None) -> None:
    n, c, h, w = dx.shape
    lib = _lib.load()
    if deterministic_enabled(deterministic):
        need = int(lib.ps_tie_workspace_ints(n, h, w))
        counts = torch.empty(need, device=dx.device, dtype=torch.int32)
        _lib.check(lib.ps_minpool_bwd_det(m.data_ptr(), arg.data_ptr(), label.data_ptr(), thr.data_ptr(), take.data_ptr(), counts.data_ptr(), need,
                                          dx.data_ptr(), grad_scale, n, c, h, w, _stream()), "ps_minpool_bwd_det")
        return
    counter = torch.zeros(n, device=dx.device, dtype=torch.int32)
    _lib.check(lib.ps_minpool_bwd(m.data_ptr(), arg.data_ptr(), label.data_ptr(), thr.data_ptr(), take.data_ptr(), counter.data_ptr(), dx.data_ptr(),
                                  grad_scale, n, c, h, w, _stream()), "ps_minpool_bwd")


# ---------------------------------------------------------------------------------------------------
# sliding-window evaluation + d4 test-time augmentation (csrc/sliding_ops.hip)
# ---------------------------------------------------------------------------------------------------
def softmax_scatter_accum(scores: Tensor, tiles_dev: Tensor, apply_softmax: bool) -> None:
    """scores [N,C,H,W] f32; tiles_dev: uint8 device tensor holding N packed `_lib.TileDst` records."""
    _require_gpu(scores, tiles_dev)
    n, c, h, w = scores.shape
    assert scores.is_contiguous() and scores.dtype == torch.float32
    assert tiles_dev.dtype == torch.uint8 and tiles_dev.numel() == n * C.sizeof(_lib.TileDst)
    lib = _lib.load()
    _lib.check(lib.ps_softmax_scatter_accum(scores.data_ptr(), n, c, h, w, tiles_dev.data_ptr(), int(apply_softmax), _stream()),
               "ps_softmax_scatter_accum")


def canvas_resize_accum(src: Tensor, src_count: Optional[Tensor], src_div: float, dst: Tensor, dst_count: Optional[Tensor], channels_last: bool,
                        zero_uncovered: bool, accumulate: bool) -> None:
    """f64 canvases [H,W,C] (channels_last) or [C,H,W]; dst (=|+=) bilinear(src / count-or-div), align_corners=False."""
    _require_gpu(src, dst)
    assert src.dtype == torch.float64 and dst.dtype == torch.float64 and src.is_contiguous() and dst.is_contiguous()
    if channels_last:
        (hs, ws, c), (hd, wd, c2) = src.shape, dst.shape
    else:
        (c, hs, ws), (c2, hd, wd) = src.shape, dst.shape
    assert c == c2
    lib = _lib.load()
    _lib.check(lib.ps_canvas_resize_accum(src.data_ptr(), _ptr(src_count), float(src_div), hs, ws, dst.data_ptr(), _ptr(dst_count), hd, wd, c,
                                          int(channels_last), int(zero_uncovered), int(accumulate), _stream()), "ps_canvas_resize_accum")


def canvas_argmax(canvas: Tensor, count: Optional[Tensor], channels_last: bool, gt: Optional[Tensor] = None, bg_value: int = -1) -> Tensor:
    _require_gpu(canvas)
    assert canvas.dtype == torch.float64 and canvas.is_contiguous()
    if channels_last:
        h, w, c = canvas.shape
    else:
        c, h, w = canvas.shape
    if gt is not None:
        assert gt.dtype == torch.uint8 and gt.is_contiguous() and tuple(gt.shape) == (h, w)
    pred = torch.empty((h, w), device=canvas.device, dtype=torch.uint8)
    lib = _lib.load()
    _lib.check(lib.ps_canvas_argmax(canvas.data_ptr(), _ptr(count), h, w, c, int(channels_last), _ptr(gt), int(bg_value), pred.data_ptr(), _stream()),
               "ps_canvas_argmax")
    return pred


def d4_view(src: Tensor, dst: Tensor, hflip: bool, k: int, inverse: bool, accumulate: bool) -> None:
    """src/dst: [..., S, S] f32 contiguous, same shape."""
    _require_gpu(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous()
    assert src.shape == dst.shape and src.shape[-1] == src.shape[-2], "d4 views need square tiles"
    s = src.shape[-1]
    lib = _lib.load()
    _lib.check(lib.ps_d4_view(src.data_ptr(), dst.data_ptr(), src.numel() // (s * s), s, int(hflip), int(k), int(inverse), int(accumulate), _stream()),
               "ps_d4_view")


def scale_inplace_(x: Tensor, divisor: float) -> None:
    _require_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    lib = _lib.load()
    _lib.check(lib.ps_scale_inplace(x.data_ptr(), x.numel(), float(divisor), _stream()), "ps_scale_inplace")
