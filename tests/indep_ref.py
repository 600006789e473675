"""Independent second formulation of the DCN ops (test infrastructure).

Written from the operator's mathematical definition, NOT from the reference's
kernels: sampling is `F.grid_sample(align_corners=True, padding_mode='zeros')` on
un-normalised coordinates, the contraction is an einsum and every gradient comes
from torch autograd.  It is used only to pin the C oracle (tests/test_oracle.py)
and to cross-check the golden fixtures (tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F


def _out_hw(H, W, kh, kw, s, p, d):
    return (H + 2 * p - (d * (kh - 1) + 1)) // s + 1, (W + 2 * p - (d * (kw - 1) + 1)) // s + 1


def sample_columns(x, offset, mask, kh, kw, stride, padding, dilation, dg):
    """-> col[N, C, K2, Ho, Wo] (sampled and, if mask is given, modulated)."""
    N, C, H, W = x.shape
    K2 = kh * kw
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    assert H > 1 and W > 1, "grid_sample normalisation needs H, W > 1"
    off = offset.reshape(N, dg, K2, 2, Ho, Wo)
    msk = mask.reshape(N, dg, K2, Ho, Wo) if mask is not None else None
    base_h = (torch.arange(Ho, dtype=x.dtype) * stride - padding).view(1, 1, Ho, 1)
    base_w = (torch.arange(Wo, dtype=x.dtype) * stride - padding).view(1, 1, 1, Wo)
    xg = x.reshape(N * dg, C // dg, H, W)
    cols = []
    for k in range(K2):
        i, j = divmod(k, kw)
        h = base_h + i * dilation + off[:, :, k, 0]
        w = base_w + j * dilation + off[:, :, k, 1]
        grid = torch.stack((2 * w / (W - 1) - 1, 2 * h / (H - 1) - 1), dim=-1).reshape(N * dg, Ho, Wo, 2)
        s = F.grid_sample(xg, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        s = s.reshape(N, dg, C // dg, Ho, Wo)
        if msk is not None:
            s = s * msk[:, :, k].unsqueeze(2)
        cols.append(s.reshape(N, C, Ho, Wo))
    return torch.stack(cols, dim=2)


def contract(col, weight, bias, groups):
    N, C, K2, Ho, Wo = col.shape
    Co = weight.shape[0]
    colg = col.reshape(N, groups, C // groups, K2, Ho * Wo)
    wg = weight.reshape(groups, Co // groups, C // groups, K2)
    out = torch.einsum("ngckp,gock->ngop", colg, wg).reshape(N, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def deform_conv(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, dg=1):
    kh, kw = weight.shape[2:]
    col = sample_columns(x, offset, mask, kh, kw, stride, padding, dilation, dg)
    return contract(col, weight, bias, groups)


def depthaware_columns(x, depth, kh, kw, stride, padding, dilation):
    N, C, H, W = x.shape
    Ho, Wo = _out_hw(H, W, kh, kw, stride, padding, dilation)
    K2 = kh * kw
    ux = F.unfold(x, (kh, kw), dilation=dilation, padding=padding, stride=stride).reshape(N, C, K2, Ho, Wo)
    ud = F.unfold(depth, (kh, kw), dilation=dilation, padding=padding, stride=stride).reshape(N, 1, K2, Ho, Wo)
    hc = torch.arange(Ho) * stride - padding + dilation * (kh - 1) // 2
    wc = torch.arange(Wo) * stride - padding + dilation * (kw - 1) // 2
    vh = (hc >= 0) & (hc < H)
    vw = (wc >= 0) & (wc < W)
    valid = (vh.view(Ho, 1) & vw.view(1, Wo)).view(1, 1, 1, Ho, Wo)
    dc = depth[:, :, hc.clamp(0, H - 1)][:, :, :, wc.clamp(0, W - 1)].reshape(N, 1, 1, Ho, Wo)
    sim = torch.exp(-(dc - ud).abs())
    sim = torch.where(valid, sim, torch.ones_like(sim))  # centre out of bounds: Di = Dval = 0
    return ux * sim


def depthaware_conv(x, depth, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    kh, kw = weight.shape[2:]
    col = depthaware_columns(x, depth.detach(), kh, kw, stride, padding, dilation)
    return contract(col, weight, bias, groups)


def deform_psroi_pool(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1,
                      part_size=None, sample_per_part=4, trans_std=0.0):
    """Deformable position-sensitive RoI pooling from its definition (Dai et al. 2017, eq. 5-6 with the
    R-FCN score-map layout): bin (i, j) of RoI n averages `sample_per_part^2` bilinear samples of score map
    (ctop, gh, gw), taken on a regular sub-grid of the bin shifted by the normalised offset of the bin's part
    cell times the RoI size.  Samples more than half a pixel outside the map are dropped, the rest are clamped
    into it.  Autograd supplies every gradient.  RoI corners must be non-negative (round = floor(x + .5)).
    Returns (output, count)."""
    part_size = out_size if part_size is None else part_size
    R = rois.shape[0]
    _, C, H, W = data.shape
    classes = 1 if no_trans else offset.shape[1] // 2
    per_class = out_channels // classes
    out = []
    cnt = torch.zeros(R, out_channels, out_size, out_size, dtype=data.dtype)
    steps = torch.arange(sample_per_part, dtype=data.dtype)
    for n in range(R):
        b = int(rois[n, 0])
        corner = torch.floor(rois[n, 1:] + 0.5)
        x_lo, y_lo = corner[0] * spatial_scale - 0.5, corner[1] * spatial_scale - 0.5
        x_hi, y_hi = (corner[2] + 1) * spatial_scale - 0.5, (corner[3] + 1) * spatial_scale - 0.5
        rw, rh = torch.clamp(x_hi - x_lo, min=0.1), torch.clamp(y_hi - y_lo, min=0.1)
        bw, bh = rw / out_size, rh / out_size
        planes = data[b].reshape(out_channels, group_size, group_size, H, W)
        rows = []
        for ctop_class in range(classes):
            maps = planes[ctop_class * per_class:(ctop_class + 1) * per_class]
            grid_out = torch.zeros(per_class, out_size, out_size, dtype=data.dtype)
            pieces = []
            for i in range(out_size):
                for j in range(out_size):
                    pi, pj = (i * part_size) // out_size, (j * part_size) // out_size
                    gi = min((i * group_size) // out_size, group_size - 1)
                    gj = min((j * group_size) // out_size, group_size - 1)
                    if no_trans:
                        tx = ty = torch.zeros((), dtype=data.dtype)
                    else:
                        tx = offset[n, 2 * ctop_class, pi, pj] * trans_std
                        ty = offset[n, 2 * ctop_class + 1, pi, pj] * trans_std
                    xs = x_lo + j * bw + tx * rw + steps * (bw / sample_per_part)
                    ys = y_lo + i * bh + ty * rh + steps * (bh / sample_per_part)
                    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
                    keep = (xx >= -0.5) & (xx <= W - 0.5) & (yy >= -0.5) & (yy <= H - 0.5)
                    k = int(keep.sum())
                    cnt[n, ctop_class * per_class:(ctop_class + 1) * per_class, i, j] = k
                    if k == 0:
                        pieces.append(torch.zeros(per_class, dtype=data.dtype))
                        continue
                    xx = torch.clamp(xx[keep], 0, W - 1)
                    yy = torch.clamp(yy[keep], 0, H - 1)
                    grid = torch.stack((2 * xx / (W - 1) - 1, 2 * yy / (H - 1) - 1), dim=-1).view(1, 1, k, 2)
                    s = F.grid_sample(maps[:, gi, gj].unsqueeze(0), grid, mode="bilinear", padding_mode="border",
                                      align_corners=True)
                    pieces.append(s.view(per_class, k).mean(dim=1))
            del grid_out
            rows.append(torch.stack(pieces, dim=1).view(per_class, out_size, out_size))
        out.append(torch.cat(rows, dim=0))
    return torch.stack(out, dim=0) if R else torch.zeros(0, out_channels, out_size, out_size, dtype=data.dtype), cnt
