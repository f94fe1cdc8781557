"""Launch sequences of bilinear_upsample2D -> conv2D 3x3 in the phase form (tfwrapper/layers.py:336-345 into :123; likelihoods.py:200-204),
shared by the engine's lowering (engine_forward / engine_backward) and the kernel tests: the matrix launches are the ordinary
phx_conv3x3_mfma_bf16 / phx_conv3x3_wgrad_mfma_bf16 entry points on other shapes, everything else is csrc/upconv.hip.

`emit(fn, *args, **kw)` records or performs one launch, `alloc(shape, dt)` gives a device buffer with `.ptr`, `alloc_zeroed(n)` an fp32
buffer that is zero when the sequence starts.  Shapes: x [B, h, w, cin] (low resolution), y_packed [B, h, w, 4 cout] = the hi-res
[B, 2h, 2w, cout] map with its pixels in (b, i, j, a, b') order."""
from . import runtime as rt

BF16 = rt.BF16


def _conv(emit, alloc, L, S, src, pack, dst, B, H, W, K, N, tag, algorithmic=True, bias_ptr=None):
    """(algorithmic=False: a frame launch -- it re-computes pixels the phase convolution also produced, so its time counts for the family and
    its FLOPs do not: the layer's algorithmic FLOPs are the phase convolution's, 18 Cin Cout per hi-res pixel)"""
    nb = int(L.conv3x3_mfma_ws_bytes(B, H, W, K, N))
    ws = alloc((max(nb // 4, 1),), rt.F32) if nb else None
    emit(L.conv3x3_mfma_bf16_ws, src.ptr, pack.ptr, dst.ptr, bias_ptr, 0, None, ws.ptr if ws is not None else None, nb, B, H, W, K, N, S,
         tag=tag, flops=18.0 * K * N * B * H * W if algorithmic else 0.0)


def _wgrad(emit, alloc, L, S, x, dy, dw_ptr, B, H, W, K, N, algorithmic=True):
    wsb = int(L.conv3x3_wgrad_ws_bytes(B, H, W, K, N))
    ws = alloc((max(wsb // 4, 1),), rt.F32)
    emit(L.conv3x3_wgrad_mfma_bf16, x.ptr, dy.ptr, dw_ptr, ws.ptr if wsb else None, wsb, B, H, W, K, N, S,
         tag="conv3x3_mfma_wgrad", flops=18.0 * K * N * B * H * W if algorithmic else 0.0)


def forward(emit, alloc, L, S, x, w_ptr, wf_w, y_packed, B, h, w, cin, cout, need_dgrad=True, bias_ptr=None):
    """y_packed <- conv3x3_SAME(resize_x2(x), W) [+ bias], frame included.  wf_w: the packed bf16 forward filter of W itself (the row
    frame's).  Returns what the backward sequence needs."""
    ef = alloc((9 * cin * 4 * cout,), BF16)
    ed = alloc((9 * cin * 4 * cout,), BF16) if need_dgrad else None
    tf = alloc((9 * cin * cout,), BF16)
    td = alloc((9 * cin * cout,), BF16) if need_dgrad else None
    b4 = alloc((4 * cout,), rt.F32) if bias_ptr is not None else None
    emit(L.upconv_pack, w_ptr, ef.ptr, ed.ptr if ed is not None else None, tf.ptr, td.ptr if td is not None else None, bias_ptr,
         b4.ptr if b4 is not None else None, cin, cout, S)
    f_rows, f_cols = alloc((6 * B, 2 * w, cin), BF16), alloc((6 * B, 2 * h, cin), BF16)
    emit(L.upconv_frame_gather, x.ptr, f_rows.ptr, f_cols.ptr, B, h, w, cin, S)
    _conv(emit, alloc, L, S, x, ef, y_packed, B, h, w, cin, 4 * cout, "conv3x3_mfma_fwd", bias_ptr=b4.ptr if b4 is not None else None)
    fr, fc = alloc((6 * B, 2 * w, cout), BF16), alloc((6 * B, 2 * h, cout), BF16)
    _conv(emit, alloc, L, S, f_rows, wf_w, fr, 1, 6 * B, 2 * w, cin, cout, "conv3x3_mfma_fwd", algorithmic=False, bias_ptr=bias_ptr)
    _conv(emit, alloc, L, S, f_cols, tf, fc, 1, 6 * B, 2 * h, cin, cout, "conv3x3_mfma_fwd", algorithmic=False, bias_ptr=bias_ptr)
    emit(L.upconv_frame_scatter, fr.ptr, fc.ptr, y_packed.ptr, B, h, w, cout, S)
    return dict(ed=ed, td=td, f_rows=f_rows, f_cols=f_cols)


def backward_prepare(emit, alloc, L, S, ctx, dy_packed, B, h, w, cout):
    """Moves the frame's gradient out of dy_packed into ctx (dy_packed is modified: its frame is zeroed).  First step of the backward pass."""
    dfr, dfc = alloc((6 * B, 2 * w, cout), BF16), alloc((6 * B, 2 * h, cout), BF16)
    emit(L.upconv_frame_gather_dy, dy_packed.ptr, dfr.ptr, dfc.ptr, B, h, w, cout, S)
    ctx["dfr"], ctx["dfc"] = dfr, dfc


def backward_filters(emit, alloc, alloc_zeroed, L, S, ctx, x, dy_packed, dw_ptr, B, h, w, cin, cout):
    """dw_hwio += the filter gradient (after backward_prepare)."""
    dfr, dfc = ctx["dfr"], ctx["dfc"]
    dweff, dwt = alloc_zeroed(9 * cin * 4 * cout), alloc_zeroed(9 * cin * cout)
    _wgrad(emit, alloc, L, S, x, dy_packed, dweff.ptr, B, h, w, cin, 4 * cout)
    _wgrad(emit, alloc, L, S, ctx["f_rows"], dfr, dw_ptr, 1, 6 * B, 2 * w, cin, cout, algorithmic=False)
    _wgrad(emit, alloc, L, S, ctx["f_cols"], dfc, dwt.ptr, 1, 6 * B, 2 * h, cin, cout, algorithmic=False)
    emit(L.upconv_fold_wgrad, dweff.ptr, dwt.ptr, dw_ptr, cin, cout, S)


def backward_data(emit, alloc, L, S, ctx, dy_packed, wd_w, dx, B, h, w, cin, cout):
    """dx [B, h, w, cin] <- the gradient with respect to the LOW-resolution input (resize adjoint included; after backward_prepare).
    wd_w: the packed bf16 data-gradient filter of W itself."""
    _conv(emit, alloc, L, S, dy_packed, ctx["ed"], dx, B, h, w, 4 * cout, cin, "conv3x3_mfma_dgrad")
    dF_rows, dF_cols = alloc((6 * B, 2 * w, cin), BF16), alloc((6 * B, 2 * h, cin), BF16)
    _conv(emit, alloc, L, S, ctx["dfr"], wd_w, dF_rows, 1, 6 * B, 2 * w, cout, cin, "conv3x3_mfma_dgrad", algorithmic=False)
    _conv(emit, alloc, L, S, ctx["dfc"], ctx["td"], dF_cols, 1, 6 * B, 2 * h, cout, cin, "conv3x3_mfma_dgrad", algorithmic=False)
    emit(L.upconv_frame_scatter_dx, dF_rows.ptr, dF_cols.ptr, dx.ptr, B, h, w, cin, S)
