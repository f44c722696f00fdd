"""CPU emulation of the *semantics* of l2i_conv2d_f32 (include/l2i.h) with torch ops — test infrastructure that
lets the host-side launch plans (weight packing, stride-2 phase decomposition, gradient plans) be checked in the
GPU-less container.  Never imported by the product."""
import torch
import torch.nn.functional as F


def emulate_launch(L, x, y, out_hw=None, in_scale=None, in_mask=None, mask=(1.0, 0.0), out_scale=None, noise=None,
                   noise_w=0.0, bias=None, residual=None, res_mask=None, out_mask=None, act=0, slope=0.2, gain=1.0, out_gain=1.0,
                   accumulate=False, tile_hint=0):
    B, cin, H, W = x.shape
    OHf, OWf = y.shape[2], y.shape[3]
    OH = (OHf - L.off_y + L.step - 1) // L.step
    OW = (OWf - L.off_x + L.step - 1) // L.step
    if OH <= 0 or OW <= 0:
        return
    w = L.w[:, :, :L.cout].reshape(L.cin, L.kh, L.kw, L.cout).permute(3, 0, 1, 2)     # unpack
    xi = x
    if in_scale is not None:
        xi = xi * in_scale.reshape(B, cin, 1, 1)
    if in_mask is not None:
        xi = xi * torch.where(in_mask > 0, torch.tensor(mask[0]), torch.tensor(mask[1]))
    # input window needed: rows o*stride - pad + k for o < OH
    need_h = (OH - 1) * L.stride + L.kh
    need_w = (OW - 1) * L.stride + L.kw
    pt, pl = max(L.pad_y, 0), max(L.pad_x, 0)
    xi = xi[:, :, max(-L.pad_y, 0):, max(-L.pad_x, 0):]
    pb = max(need_h - pt - xi.shape[2], 0)
    pr = max(need_w - pl - xi.shape[3], 0)
    xi = F.pad(xi, [pl, pr, pt, pb])[:, :, :need_h, :need_w]
    a = F.conv2d(xi, w, stride=L.stride)
    assert a.shape[2] == OH and a.shape[3] == OW, (a.shape, OH, OW)
    if out_scale is not None:
        a = a * out_scale.reshape(B, L.cout, 1, 1)
    sl = (slice(None), slice(None), slice(L.off_y, None, L.step), slice(L.off_x, None, L.step))
    if out_mask is not None:
        a = torch.where(out_mask[sl] > 0, a, torch.zeros_like(a))
    if noise is not None:
        a = a + noise[sl] * noise_w
    if bias is not None:
        a = a + bias.reshape(1, -1, 1, 1)
    if residual is not None:
        r = residual[sl]
        if res_mask is not None:
            r = torch.where(res_mask[sl] > 0, r, torch.zeros_like(r))
        a = a + r
    if act == 1:
        a = torch.where(a > 0, a, a * slope) * gain
    elif act == 2:
        a = torch.relu(a)
    a = a * out_gain
    if accumulate:
        a = a + y[sl]
    y[sl] = a


def emulate_fused_transposed(Fz, x, y, in_scale=None, in_mask=None, mask=(1.0, 0.0), out_scale=None, out_gain=1.0, tile_hint=0):
    """Semantics of l2i_conv_transpose2d_f32: unpack the fused tap order and run torch's conv_transpose2d."""
    from latent2im_amd import conv
    B, cin, H, W = x.shape
    K = Fz.k
    taps = conv.fused_transposed_taps(K, Fz.pad)
    w = torch.zeros(Fz.cout, cin, K, K)
    packed = Fz.w[:, :, :Fz.cout]                                   # [Cin, K*K, Cout]
    for s, (ky, kx) in enumerate(taps):
        w[:, :, ky, kx] = packed[:, s, :].t()
    xi = x
    if in_scale is not None:
        xi = xi * in_scale.reshape(B, cin, 1, 1)
    if in_mask is not None:
        xi = xi * torch.where(in_mask > 0, torch.tensor(mask[0]), torch.tensor(mask[1]))
    nat_h, nat_w = (H - 1) * 2 - 2 * Fz.pad + K, (W - 1) * 2 - 2 * Fz.pad + K
    out = F.conv_transpose2d(xi, w.transpose(0, 1), stride=2, padding=Fz.pad,
                             output_padding=(y.shape[2] - nat_h, y.shape[3] - nat_w))
    if out_scale is not None:
        out = out * out_scale.reshape(B, Fz.cout, 1, 1)
    y.copy_(out * out_gain)
    return y


def wino_conv(x, U, cout, pad):
    """What l2i_conv2d_wino_f32 computes before its epilogue, from the packed U = G g G^T [Cin,4,CoutP,4]: per 2x2 output
    tile  Y = A^T [ sum_c U[c] . (B^T d_c B) ] A  (zero padding, odd sizes via overhanging tiles)."""
    Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=x.dtype)
    At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=x.dtype)
    n, cin, H, W = x.shape
    OH, OW = H + 2 * pad - 2, W + 2 * pad - 2
    ty, tx = (OH + 1) // 2, (OW + 1) // 2
    xp = F.pad(x, [pad, 2 * tx + 2 - W - pad, pad, 2 * ty + 2 - H - pad])
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                             # [n, cin, ty, tx, 4, 4]
    V = torch.einsum('ik,nctukl,jl->nctuij', Bt, d, Bt)
    M = torch.einsum('cioj,nctuij->notuij', U.to(x.dtype)[:, :, :cout, :], V)
    Y = torch.einsum('ai,notuij,bj->notaub', At, M, At)               # [n, cout, ty, 2, tx, 2]
    return Y.reshape(n, cout, 2 * ty, 2 * tx)[:, :, :OH, :OW]


def seeded_state(module, seed, scale=1.0):
    """Same rule as tests/golden/make_golden.py::seeded_module_state: RandomState(seed).standard_normal per tensor in
    state_dict order, matrices * scale / sqrt(fan_in), vectors * 0.1.  Loads the values into ``module`` and returns them."""
    import numpy as np
    import torch
    rs = np.random.RandomState(seed)
    sd = {}
    for k, v in module.state_dict().items():
        a = rs.standard_normal(tuple(v.shape)).astype(np.float32)
        sd[k] = (a * (scale / np.sqrt(v.shape[1])) if v.dim() == 2 else a * 0.1).astype(np.float32)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd


def emulate_segmented_matvec(out, inp, w, segs, block_seg, nblocks, B, in2=None, bias=None, e1=None, e2=None, wmod=None, wrgb=None):
    """Semantics of l2i_segmented_matvec_f32 (include/l2i.h) in float64 numpy, driven by the same device-format segment tables the product
    hands the kernel (latent2im_amd/generator.py:_ModPlan): walks the (segment, row block) list block by block like the grid does."""
    import ctypes
    import numpy as np
    from latent2im_amd import _lib
    raw = bytes(segs.cpu().numpy().tobytes())
    n = len(raw) // ctypes.sizeof(_lib.SegmvSeg)
    table = (_lib.SegmvSeg * n).from_buffer_copy(raw)
    bs = block_seg.cpu().numpy().reshape(-1, 2)
    assert len(bs) == nblocks
    f = lambda t: None if t is None else t.detach().cpu().double().numpy().reshape(-1)
    o, i1, i2, W, bi, E1, E2, wm, wr = out.detach().cpu().double().numpy().reshape(-1).copy(), f(inp), f(in2), f(w), f(bias), f(e1), f(e2), \
        (None if wmod is None else wmod.detach().cpu().double().numpy().reshape(-1).copy()), f(wrgb)
    written = np.zeros(o.shape, dtype=bool)
    for si, rb in bs:
        sg = table[si]
        r = np.arange(rb * 64, min(rb * 64 + 64, sg.rows))
        for b in range(B):
            acc = np.zeros(len(r))
            for pi in range(sg.nparts):
                pt = sg.part[pi]
                assert pt.K % 4 == 0 and pt.K <= 512
                base = pt.in_off_c + B * pt.in_off_b
                k = np.arange(pt.K)
                if pt.pre == 3:
                    x = sum(i2[base + (b * pt.K + k) * 3 + q] * wr[pt.aux_off + q * pt.K + k] for q in range(3))
                else:
                    x = i1[base + b * pt.in_bstride + k]
                    if pt.pre == 1:
                        x = x * x
                    elif pt.pre == 2:
                        x = x * i2[base + b * pt.in_bstride + k] ** 2
                Wm = W[pt.w_off + k[:, None] * pt.w_pitch + r[None, :]]
                acc = acc + x @ Wm
            v = acc
            if sg.epi == 0:
                v = v + bi[sg.bias_off + r]
            elif sg.epi == 1:
                v = 1.0 / np.sqrt(v + 1e-8)
            elif sg.epi == 2:
                ei = sg.e_off_c + B * sg.e_off_b + b * sg.e_bstride + r
                v = E1[ei] - E2[ei] * v
            idx = sg.out_off_c + B * sg.out_off_b + b * sg.out_bstride + r
            assert not written[idx].any(), 'two blocks write the same output'
            written[idx] = True
            o[idx] = v
            if sg.epi == 0 and sg.rgb_off_b >= 0:
                for q in range(3):
                    wm[B * sg.rgb_off_b + b * 3 * sg.rows + q * sg.rows + r] = wr[sg.rgb_w_off + q * sg.rows + r] * v
    assert written.all(), 'output elements left unwritten'
    out.copy_(torch.from_numpy(o).to(out.dtype).reshape(out.shape))
    if wmod is not None:
        wmod.copy_(torch.from_numpy(wm).to(wmod.dtype).reshape(wmod.shape))
    return out
