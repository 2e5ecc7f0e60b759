"""The algebra behind the MFMA path of ExtractorAttn's FC layer (csrc/fc_gemm.hip), checked on the host in float64
with the geometry the library itself reports (gfla_fc_geometry) -- no GPU involved:

  1. "sample the convolved map": conv_{k, stride k}(block_extractor(source, flow), W) equals the bilinear sample, at
     p + flow(p), of conv_kxk(replicate-extended source, W) with the reference's clamped-index / unclamped-weight
     corners (block_extractor_kernel.cu:66-76), using the oracle's literal block_extractor as the left-hand side;
  2. the linearised layouts the kernels index: a k x k tap is the constant pixel offset i*Wp + j; the gradient map in
     "Z layout" (k-1 zero columns per row, (k-1)*(Wp+1) leading zeros) turns the transposed convolution into the same
     linear form with flipped taps, and the weight gradient into a pixel reduction against the shifted input;
  3. folding the replicate padding back.
"""
import torch
import torch.nn.functional as F

from util import make_flow, randn


def _geometry(H, W, k, is_source):
    from global_flow_local_attention_amd import fc_mfma
    return fc_mfma.geometry(H, W, k, is_source)


def _corners(flow, H, W, k, wps):
    """fc_sample.hip::corners in torch: indices into the convolved map (row pitch wps) and the four weights."""
    lo, hi = k // 2, k - 1 - k // 2
    B = flow.size(0)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    dx = flow[:, 0] + xs.to(flow.dtype)
    dy = flow[:, 1] + ys.to(flow.dtype)
    fdx, fdy = torch.floor(dx), torch.floor(dy)
    xr, yb = dx - fdx, dy - fdy
    xl, yt = 1 - xr, 1 - yb
    qx, qy = fdx.long(), fdy.long()
    gx0 = qx.clamp(-hi, W - 1 + lo) + hi
    gx1 = (qx + 1).clamp(-hi, W - 1 + lo) + hi
    gy0 = qy.clamp(-hi, H - 1 + lo) + hi
    gy1 = (qy + 1).clamp(-hi, H - 1 + lo) + hi
    idx = [gy0 * wps + gx0, gy0 * wps + gx1, gy1 * wps + gx0, gy1 * wps + gx1]
    wts = [xl * yt, xr * yt, xl * yb, xr * yb]
    return [i.reshape(B, -1) for i in idx], [w.reshape(B, -1) for w in wts]


def test_sampling_the_convolved_map_equals_convolving_the_samples(oracle):
    from oracle import cpu_modules
    for k, kind in ((3, "coherent"), (5, "wild"), (4, "smooth"), (3, "integer"), (5, "zero")):
        B, C, H, W, N = 2, 5, 9, 7, 6
        s = randn((B, C, H, W), torch.float64, seed=1)
        f = make_flow(kind, B, H, W, torch.float64, seed=2)
        w = randn((N, C, k, k), torch.float64, seed=3)
        lhs = F.conv2d(cpu_modules._BlockExtractorCPU.apply(s, f, k), w, stride=k)      # the reference's form
        g = _geometry(H, W, k, True)
        G = F.conv2d(F.pad(s, (g["pad_l"], k - 1, g["pad_t"], k - 1), mode="replicate"), w)  # (B,N,Ho,Wo)
        assert G.shape[2:] == (g["Ho"], g["Wo"])
        Glin = torch.zeros(B, N, g["Ho"] * g["Wp"], dtype=torch.float64)
        Glin.view(B, N, g["Ho"], g["Wp"])[..., :g["Wo"]] = G
        idx, wts = _corners(f, H, W, k, g["Wp"])
        rhs = sum(wt[:, None, :] * torch.gather(Glin, 2, ix[:, None, :].expand(B, N, -1)) for ix, wt in zip(idx, wts))
        assert (lhs.reshape(B, N, -1) - rhs).abs().max().item() < 1e-12, (k, kind)


def test_linearised_convolution_data_and_weight_gradients():
    for k, is_source in ((3, 1), (5, 1), (3, 0), (5, 0), (4, 1)):
        B, C, H, W, N = 1, 3, 6, 5, 4
        g = _geometry(H, W, k, is_source)
        Wp, lead = g["Wp"], g["lead"]
        x = randn((B, C, H, W), torch.float64, seed=4).requires_grad_()
        w = randn((N, C, k, k), torch.float64, seed=5).requires_grad_()
        pads = (g["pad_l"], g["Wp"] - W - g["pad_l"], g["pad_t"], g["Hp"] - H - g["pad_t"])
        xp = F.pad(x, pads, mode="replicate")
        G = F.conv2d(xp, w)
        dG = randn(tuple(G.shape), torch.float64, seed=6)
        G.backward(dG)
        # forward, linear form: out[m] = sum_{i,j,c} X[m + i*Wp + j, c] * w[n, c, i, j]
        X = torch.zeros(int(g["Sx"]), C, dtype=torch.float64)
        X[:g["Hp"] * Wp] = xp.detach()[0].permute(1, 2, 0).reshape(-1, C)
        M = g["M"]
        out = torch.zeros(M, N, dtype=torch.float64)
        for i in range(k):
            for j in range(k):
                out += X[i * Wp + j:i * Wp + j + M] @ w.detach()[:, :, i, j].t()
        rows = (torch.arange(g["Ho"])[:, None] * Wp + torch.arange(g["Wo"])[None, :]).reshape(-1)
        assert (out[rows].t().reshape(N, g["Ho"], g["Wo"]) - G.detach()[0]).abs().max().item() < 1e-12
        # gradient map in Z layout
        Z = torch.zeros(int(g["Sz"]), N, dtype=torch.float64)
        Z[lead + rows] = dG[0].permute(1, 2, 0).reshape(-1, N)
        # data gradient: the same linear form with flipped taps and swapped channel roles, over the padded domain
        Md = g["Md"]
        dxp = torch.zeros(Md, C, dtype=torch.float64)
        for i in range(k):
            for j in range(k):
                dxp += Z[i * Wp + j:i * Wp + j + Md] @ w.detach()[:, :, k - 1 - i, k - 1 - j]
        dxp = dxp.reshape(g["Hp"], Wp, C)
        # fold the replicate padding (fc_sample.hip::fc_fold_kernel)
        gx = torch.zeros(C, H, W, dtype=torch.float64)
        for y in range(H):
            y0 = 0 if y == 0 else y + g["pad_t"]
            y1 = g["Hp"] - 1 if y == H - 1 else y + g["pad_t"]
            for xx in range(W):
                x0 = 0 if xx == 0 else xx + g["pad_l"]
                x1 = Wp - 1 if xx == W - 1 else xx + g["pad_l"]
                gx[:, y, xx] = dxp[y0:y1 + 1, x0:x1 + 1].sum((0, 1))
        assert (gx - x.grad[0]).abs().max().item() < 1e-12, (k, is_source)
        # weight gradient: dW[n, c, i, j] = sum_m X[m + i*Wp + j, c] * Z[lead + m, n]
        for i in range(k):
            for j in range(k):
                dw = Z[lead:lead + M].t() @ X[i * Wp + j:i * Wp + j + M]
                assert (dw - w.grad[:, :, i, j]).abs().max().item() < 1e-12, (k, is_source, i, j)


def test_geometry_has_the_slack_the_kernels_read():
    for (H, W, k) in ((64, 44, 5), (32, 22, 3), (7, 5, 3), (1, 1, 5), (64, 64, 5)):
        for src in (0, 1):
            g = _geometry(H, W, k, src)
            halo = (k - 1) * (g["Wp"] + 1)
            assert g["lead"] == halo and g["Mg"] >= g["Ho"] * g["Wo"] and g["Mdg"] >= g["Md"]
            # the convolution's largest row tile (256 outputs) starts at most at the last valid output's pixel and
            # reads its rows' span (256 + one wrap of k-1 pixels per row crossed) + the tap halo
            span = 256 + (255 // g["Wo"] + 1) * (k - 1)
            last = (g["Ho"] - 1) * g["Wp"] + g["Wo"] - 1
            assert last + halo + 1 == g["Md"] and g["Sx"] >= last + span + halo
            assert g["Sz"] >= g["Md"] + 256 + halo      # the transposed convolution's last input tile
            assert g["Sz"] >= g["lead"] + (g["M"] + 63) // 64 * 64   # the weight gradient's last K step
            assert g["Ho"] == (H + k - 1 if src else H) and g["Wo"] == (W + k - 1 if src else W)
