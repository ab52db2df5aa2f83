"""Host-side packing of the fused MLP kernels (pn2_sa_mlp3_pack, pn2_fp_mlp_pack) against a numpy emulation of
the kernels' MFMA data flow (CPU; no device work). The kernels compute fp32 products on the bf16 matrix pipe:
every weight is stored as three bf16 levels w0 + w1 + w2 (exactly w), every activation is split the same way
in registers, and six v_mfma_f32_32x32x16_bf16 per 16 contraction slots add the terms listed in _TERMS
(csrc/sa_mlp_common.h, mma_x6). The emulation hard-codes the operand maps of that instruction -- A: lane l
holds A[i = l & 31][k = 8 (l >> 5) + j], j = 0..7; B: B[k = 8 (l >> 5) + j][l & 31]; C/D: lane l, register v
holds D[8(v >> 2) + 4(l >> 5) + (v & 3)][l & 31] -- and walks the packed arrays exactly as the kernels do, so
it proves the permutation algebra (slot <-> channel map, level placement, bias, layer chaining, stream order)
GIVEN those maps; the maps themselves and the accuracy are checked on the GPU (tests/test_sa_mlp_gpu.py).
To make every one of the six terms visible, the emulated activations are "split" into the UNEQUAL parts
_COEF * x instead of bf16 levels: the expected result is then the layer stack with the effective weights
w0 (a + b + c) + w1 (a + b) + w2 a, where w0, w1, w2 come from an independent numpy bf16 split."""
import ctypes

import numpy as np
import pytest

_TERMS = [(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]          # (weight level, activation level)
_COEF = (0.5, 0.3, 0.2)
_PAIR = 1536                                                        # 4-byte words per 32x32 tile pair


def _chan(v, h):
    return 8 * (v >> 2) + 4 * h + (v & 3)


_L = np.arange(64)
_ROW = np.array([[8 * (v >> 2) + 4 * (l >> 5) + (v & 3) for v in range(16)] for l in range(64)])   # C/D row of (lane, reg)
_K = (8 * (_L >> 5))[:, None] + np.arange(8)[None, :]                                              # k of (lane, slot)


def _mfma(a, b, acc):
    """One v_mfma_f32_32x32x16_bf16: a, b (64 lanes, 8 slots), acc (64, 16) -> acc + A.B in the C/D map."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    A[(_L & 31)[:, None], _K] = a
    B[_K, (_L & 31)[:, None]] = b
    D = A @ B
    return acc + D[_ROW, (_L & 31)[:, None]]


def _bf16_bits_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def _bf16_split(w):
    """Independent restatement of the host split: three round-to-nearest-even bf16 levels of float32 values."""
    levels = []
    r = np.asarray(w, np.float32)
    for _ in range(3):
        u = r.view(np.uint32).astype(np.uint64)
        hi = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
        levels.append(hi.astype(np.float64))
        r = (r - hi).astype(np.float32)
    return levels


def _effective(w):
    w0, w1, w2 = _bf16_split(w)
    a, b, c = _COEF
    return w0 * (a + b + c) + w1 * (a + b) + w2 * a


def _emulate_pair(pair, act, acc, last):
    """The 12 MFMAs of one packed tile pair ([e][level][lane][8 bf16]) on activation registers act (64, 16);
    last: operands swapped."""
    W = _bf16_bits_to_f64(pair.view(np.uint16).reshape(2, 3, 64, 8))
    for e in range(2):
        X = [c * act[:, 8 * e:8 * e + 8] for c in _COEF]
        for wl, xl in _TERMS:
            acc = _mfma(X[xl], W[e, wl], acc) if last else _mfma(W[e, wl], X[xl], acc)
    return acc


def _want(x, ws, bs):
    want = x.astype(np.float64)
    for w, b in zip(ws, bs):
        want = np.maximum(want @ _effective(w) + b, 0.0)
    return want


def test_weight_levels_are_exact():
    """w0 + w1 + w2 == w exactly (three nearest-even bf16 roundings cover fp32's 24 bits) and the packed levels of a
    tile pair are the numpy split's."""
    from pointnet2_amd import _C
    lib = _C.lib()
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((32, 32)) * np.exp(rng.uniform(-20, 20, (32, 32)))).astype(np.float32)
    lv = _bf16_split(w)
    assert np.array_equal((lv[0] + lv[1] + lv[2]).astype(np.float32), w)
    info = (ctypes.c_int * 4)()
    wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
    zeros = np.zeros(32, np.float32)
    assert lib.pn2_sa_mlp3_config(32, 32, 32, 32, 32, info, ctypes.byref(wf), ctypes.byref(bf)) == 0 and info[0] == 0
    wp, bp = np.empty(wf.value, np.float32), np.empty(bf.value, np.float32)
    assert lib.pn2_sa_mlp3_pack(32, 32, 32, 32, 32, 1, w.ctypes.data, zeros.ctypes.data, w.ctypes.data, zeros.ctypes.data,
                                w.ctypes.data, zeros.ctypes.data, wp.ctypes.data, bp.ctypes.data) == 0
    pair = _bf16_bits_to_f64(wp[_PAIR:2 * _PAIR].view(np.uint16).reshape(2, 3, 64, 8))          # layer 2: no row permutation
    for e in range(2):
        for lane in range(64):
            for j in range(8):
                k, n = _chan(8 * e + j, lane >> 5), lane & 31
                assert [pair[e, l, lane, j] for l in range(3)] == [lv[l][k, n] for l in range(3)]


def _emulate_layer(wp, bp, t_out, t_in, acts, last=False):
    """acts: list of t_in arrays (64 lanes, 16 regs) -> list of t_out arrays, exactly as mlp_layer walks them.
    last: the kernel swaps the MFMA operands and starts from zero (bias + ReLU come after the pooling)."""
    wp = wp.reshape(t_out, t_in, _PAIR)
    bp = bp.reshape(t_out, 2, 16)
    outs = []
    for t in range(t_out):
        acc = np.zeros((64, 16)) if last else np.stack([bp[t, l >> 5] for l in range(64)]).astype(np.float64)
        for u in range(t_in):
            acc = _emulate_pair(wp[t, u], acts[u], acc, last)
        outs.append(acc if last else np.maximum(acc, 0.0))
    return outs


@pytest.mark.parametrize("cin,widths", [(3, (32, 32, 64)), (6, (64, 64, 128)), (9, (64, 96, 128)), (3, (16, 20, 40))])
def test_pack_matches_emulated_dataflow(cin, widths):
    from pointnet2_amd import _C
    lib = _C.lib()
    rng = np.random.default_rng(5)
    dims = (cin,) + tuple(widths)
    ws = [rng.standard_normal((dims[i], dims[i + 1])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(3)]
    tiles = (ctypes.c_int * 4)()
    wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
    assert lib.pn2_sa_mlp3_config(cin, *widths, 32, tiles, ctypes.byref(wf), ctypes.byref(bf)) == 0
    assert tiles[0] == 0                                            # resident kernel
    _, t1, t2, t3 = tiles
    wp = np.empty(wf.value, np.float32)
    bp = np.empty(bf.value, np.float32)
    assert lib.pn2_sa_mlp3_pack(cin, *widths, 32, 1, *[a.ctypes.data for pair in zip(ws, bs) for a in pair], wp.ctypes.data,
                                bp.ctypes.data) == 0
    x = rng.standard_normal((32, cin)).astype(np.float32)          # 32 samples
    # layer-1 operand registers: register v of lane l = input channel chan(v, l >> 5) of sample l & 31
    x0 = np.zeros((64, 16))
    for l in range(64):
        for v in range(16):
            k = _chan(v, l >> 5)
            x0[l, v] = x[l & 31, k] if k < cin else 0.0
    sizes_w = [t1 * 1 * _PAIR, t2 * t1 * _PAIR, t3 * t2 * _PAIR]
    sizes_b = [t1 * 32, t2 * 32, t3 * 32]
    ow = np.cumsum([0] + sizes_w)
    ob = np.cumsum([0] + sizes_b)
    acts = [x0]
    for L, (to, ti) in enumerate([(t1, 1), (t2, t1), (t3, t2)]):
        acts = _emulate_layer(wp[ow[L]:ow[L + 1]], bp[ob[L]:ob[L + 1]], to, ti, acts, last=(L == 2))
    # the last layer's registers: lane l holds channel 32t + (l & 31) of sample 8(v >> 2) + 4(l >> 5) + (v & 3);
    # the bias comes from the packed array through the kernel's inverse map (b3_at)
    b3 = bp[ob[2]:ob[3]].reshape(t3, 2, 16)
    got = np.zeros((32, widths[2]))
    for t in range(t3):
        for l in range(64):
            ch = 32 * t + (l & 31)
            if ch >= widths[2]:
                continue
            c = ch & 31
            bias = b3[t, (c >> 2) & 1, 4 * (c >> 3) + (c & 3)]
            for v in range(16):
                got[8 * (v >> 2) + 4 * (l >> 5) + (v & 3), ch] = max(acts[t][l, v] + bias, 0.0)
    assert np.allclose(got, _want(x, ws, bs), rtol=1e-9, atol=1e-9)


def test_config_limits():
    from pointnet2_amd import _C
    lib = _C.lib()
    info = (ctypes.c_int * 4)()
    assert lib.pn2_sa_mlp3_config(3, 64, 64, 128, 32, info, None, None) == 0 and info[0] == 0
    assert lib.pn2_sa_mlp3_config(67, 64, 64, 128, 32, info, None, None) == 0 and info[0] == 1    # wide input: streamed
    assert lib.pn2_sa_mlp3_config(131, 128, 128, 256, 64, info, None, None) == 0 and list(info) == [1, 4, 4, 8]
    # nsample not a multiple of 32 / wide stacks: the cooperative kernel (masked tail, output tiles split over the waves)
    assert lib.pn2_sa_mlp3_config(131, 128, 128, 256, 16, info, None, None) == 0 and list(info) == [2, 4, 4, 8]
    assert lib.pn2_sa_mlp3_config(259, 256, 256, 512, 32, info, None, None) == 0 and list(info) == [2, 8, 8, 16]   # sem_seg SA4
    assert lib.pn2_sa_mlp3_config(259, 256, 512, 1024, 128, info, None, None) == 0 and list(info) == [2, 8, 16, 32]  # group_all
    assert lib.pn2_sa_mlp3_config(643, 256, 512, 1024, 100, info, None, None) == 0 and info[0] == 2                 # any nsample
    assert lib.pn2_sa_mlp3_config(259, 512, 512, 1024, 32, info, None, None) != 0   # hidden layers beyond the LDS exchange
    assert lib.pn2_sa_mlp3_config(2, 64, 64, 128, 32, info, None, None) != 0
    assert lib.pn2_sa_mlp3_config(3, 64, 64, 128, 48, info, None, None) != 0


def _operand_tiles(x, ti):
    """(32 samples, cin) -> ti operand tiles (64 lanes, 16 regs): register v of lane l = channel 32u + chan(v, l >> 5)."""
    tiles = []
    for u in range(ti):
        t = np.zeros((64, 16))
        for l in range(64):
            for v in range(16):
                k = 32 * u + _chan(v, l >> 5)
                t[l, v] = x[l & 31, k] if k < x.shape[1] else 0.0
        tiles.append(t)
    return tiles


def _unswap(acc_tiles, cout, bias_of):
    """Last-layer accumulators (swapped operands: lane = channel, register = sample) -> (32, cout) with bias + ReLU."""
    got = np.zeros((32, cout))
    for t, acc in acc_tiles.items():
        for l in range(64):
            ch = 32 * t + (l & 31)
            if ch < cout:
                for v in range(16):
                    got[_chan(v, l >> 5), ch] = max(acc[l, v] + bias_of(ch), 0.0)
    return got


def _b_at(bp_layer, ch):
    t, c = ch >> 5, ch & 31
    return bp_layer.reshape(-1, 2, 16)[t, (c >> 2) & 1, 4 * (c >> 3) + (c & 3)]


@pytest.mark.parametrize("cin,widths,xyz_first", [(67, (64, 64, 128), True), (131, (128, 128, 256), True),
                                                   (40, (50, 64, 100), False), (323, (128, 128, 256), False),
                                                   (3, (128, 128, 256), True)])
def test_stream_pack_matches_emulated_dataflow(cin, widths, xyz_first):
    """The streamed kernels' weight sequences, emulated exactly as sa_mlp_stream.hip walks them: the grouped kernel's
    stream = [layer 2 input-tile-major, layer 3 output-tile-major]; behind it the per-point kernel's stream = the
    FEATURE rows of layer 1, input-tile-major, padded to a stage; last the xyz rows of layer 1 (K16 step 0 of one
    pair per output tile, LDS-resident). Layer 1 = P[j] (features . W1f + b1, once per point) + xyz part."""
    from pointnet2_amd import _C
    lib = _C.lib()
    rng = np.random.default_rng(cin)
    dims = (cin,) + tuple(widths)
    ws = [rng.standard_normal((dims[i], dims[i + 1])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(3)]
    info = (ctypes.c_int * 4)()
    wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
    assert lib.pn2_sa_mlp3_config(cin, *widths, 32, info, ctypes.byref(wf), ctypes.byref(bf)) == 0
    assert info[0] == 1
    _, t1, t2, t3 = info
    cfeat = cin - 3
    tif = (cfeat + 31) // 32
    wp = np.empty(wf.value, np.float32)
    bp = np.empty(bf.value, np.float32)
    assert lib.pn2_sa_mlp3_pack(cin, *widths, 32, 1 if xyz_first else 0,
                                *[a.ctypes.data for pair in zip(ws, bs) for a in pair], wp.ctypes.data, bp.ctypes.data) == 0
    pad = lambda k: -(-k // 4) * 4
    main = t2 * t1 + t3 * t2
    nstream = main + pad(tif * t1)
    assert wp.size == nstream * _PAIR + t1 * (_PAIR // 2)
    pairs = wp[:nstream * _PAIR].reshape(-1, _PAIR)
    xyz_half = wp[nstream * _PAIR:].reshape(t1, _PAIR // 2)
    bias = bp.reshape(-1, 2, 16)
    xyz = rng.standard_normal((32, 3))
    feat = rng.standard_normal((32, cfeat))
    user_in = np.concatenate([xyz, feat], axis=1) if xyz_first else np.concatenate([feat, xyz], axis=1)

    lane_bias = lambda t0, t: np.stack([bias[t0 + t, l >> 5] for l in range(64)]).astype(np.float64)
    # per-point kernel: P = features . W1f + b1 (non-swapped: lane = point, register v = channel chan(v, h))
    h1 = [lane_bias(0, t) for t in range(t1)]
    for u, x0 in enumerate(_operand_tiles(feat, tif)):
        for t in range(t1):
            h1[t] = _emulate_pair(pairs[main + u * t1 + t], x0, h1[t], False)
    # grouped kernel, layer 1: + xyz part, K16 step 0 of the xyz pairs only (step 1 must be all zero)
    x0 = _operand_tiles(xyz, 1)[0]
    for t in range(t1):
        full = np.concatenate([xyz_half[t], np.zeros(_PAIR // 2, np.float32)])      # K16 step 1 is not even stored
        h1[t] = _emulate_pair(full, x0, h1[t], False)
    h1 = [np.maximum(a, 0.0) for a in h1]
    l1 = 0
    h2 = [lane_bias(t1, t) for t in range(t2)]
    for u in range(t1):                                              # input tiles outermost
        for t in range(t2):
            h2[t] = _emulate_pair(pairs[l1 + u * t2 + t], h1[u], h2[t], False)
    h2 = [np.maximum(a, 0.0) for a in h2]
    b3 = bp[(t1 + t2) * 32:]
    acc = {t: np.zeros((64, 16)) for t in range(t3)}
    for t in range(t3):
        for u in range(t2):
            acc[t] = _emulate_pair(pairs[l1 + t2 * t1 + t * t2 + u], h2[u], acc[t], True)
    got = _unswap(acc, widths[2], lambda ch: _b_at(b3, ch))
    assert np.allclose(got, _want(user_in, ws, bs), rtol=1e-9, atol=1e-9)


def test_fold_batch_norm_equals_eval_mode_layers():
    """sa_mlp.fold_batch_norm against torch's own conv 1x1 + BatchNorm2d (eval) on the CPU, and
    _SharedMLP.folded_layers() end to end (three layers + ReLU)."""
    import torch
    from pointnet2_amd import sa_mlp
    from pointnet2_amd.pointnet_util import _SharedMLP
    torch.manual_seed(3)
    mlp = _SharedMLP(7, [16, 12, 20], bn=True)
    for mod in mlp.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.uniform_(-0.5, 0.5)
            mod.running_var.uniform_(0.3, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
    mlp.eval()
    x = torch.randn(2, 7, 5, 6)
    with torch.no_grad():
        want = mlp(x)
    layers = mlp.folded_layers()
    assert [w.shape for w, _ in layers] == [(7, 16), (16, 12), (12, 20)]
    act = x.permute(0, 2, 3, 1).double().numpy()
    for w, b in layers:
        act = np.maximum(act @ w.astype(np.float64) + b.astype(np.float64), 0.0)
    assert np.allclose(act, want.permute(0, 2, 3, 1).double().numpy(), rtol=1e-5, atol=1e-5)
    # without batch norm the fold is the identity on (W^T, b)
    conv = torch.nn.Conv2d(4, 6, 1)
    w, b = sa_mlp.fold_batch_norm(conv.weight, conv.bias, None)
    assert np.allclose(w, conv.weight.detach().numpy()[:, :, 0, 0].T) and np.allclose(b, conv.bias.detach().numpy())


# ---- packing of the round-2 kernels: feature propagation (streamed, kind 0) and cooperative (kind 1) ---------------
@pytest.mark.parametrize("kind", [0, 1], ids=["streamed", "cooperative"])
@pytest.mark.parametrize("c2,c1,widths", [(40, 8, (100, 60)), (24, 0, (128, 128, 70)), (36, 5, (130, 128))])
def test_fp_pack_matches_emulated_dataflow(kind, c2, c1, widths):
    """pn2_fp_mlp_pack: walk the packed streams exactly as the kernels do and compare with the plain layer stack.
    Layer 1 = bias + interp(points2 . W1a) + W1b^T points1: the known-feature rows of W1 go to the per-point kernel's
    streams (one per 128 output channels, feature tiles outermost) behind the main stream, which holds the skip-link
    rows of layer 1 and the later layers (fp_mlp.hip: one wave, input tiles outermost in layer 1 and padded to a
    stage, output tiles outermost afterwards; coop_mlp.hip: pair 4k + w belongs to wave w, input tiles outermost in
    every layer, wave w owns output tiles 4g + w). The interpolation is linear, so the emulation feeds the per-point
    kernel the already interpolated features of the 32 points."""
    from pointnet2_amd import _C
    lib = _C.lib()
    rng = np.random.default_rng(c2 + kind)
    n = len(widths)
    warr = (ctypes.c_int * n)(*widths)
    tiles = (ctypes.c_int * 4)()
    wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
    rc = lib.pn2_fp_mlp_config(c2, c1, n, warr, kind, tiles, ctypes.byref(wf), ctypes.byref(bf))
    if rc != 0:
        pytest.skip("no kernel of this kind for the stack")
    ti, T = tiles[0], [tiles[1], tiles[2], tiles[3]]
    assert ti == (c1 + 31) // 32
    tif = (c2 + 31) // 32
    dims = (c2 + c1,) + tuple(widths)
    ws = [rng.standard_normal((dims[i], dims[i + 1])).astype(np.float32) for i in range(n)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(n)]
    wp = np.empty(wf.value, np.float32)
    bp = np.empty(bf.value, np.float32)
    wptr = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
    bptr = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    assert lib.pn2_fp_mlp_pack(c2, c1, n, warr, kind, wptr, bptr, wp.ctypes.data, bp.ctypes.data) == 0
    pairs = wp.reshape(-1, _PAIR)
    ob = np.cumsum([0, T[0] * 32, T[1] * 32, T[2] * 32])
    bl = [bp[ob[i]:ob[i + 1]] for i in range(3)]
    assert bp.size == ob[3] + 128 and not bp[ob[3]:].any()            # the per-point kernel's zero bias
    x = rng.standard_normal((32, c2 + c1)).astype(np.float32)
    npoint = tif * T[0]
    main = pairs.shape[0] - npoint
    lane_bias = lambda L, t: np.stack([bl[L].reshape(-1, 2, 16)[t, l >> 5] for l in range(64)]).astype(np.float64)
    # per-point kernel on the interpolated features: stream `half` holds output tiles 4 half .. 4 half + 3
    q = {t: np.zeros((64, 16)) for t in range(T[0])}
    for u, xt in enumerate(_operand_tiles(x[:, :c2], tif)):
        for t in range(T[0]):
            half = t // 4
            q[t] = _emulate_pair(pairs[main + (half * tif + u) * 4 + t % 4], xt, q[t], False)
    skip = _operand_tiles(x[:, c2:], ti)
    tin = [ti, T[0], T[1]]
    last_acc = {}
    k = 0
    acts = skip
    for L in range(n):
        last = L == n - 1
        acc = {t: (np.zeros((64, 16)) if last else lane_bias(L, t)) for t in range(T[L])}
        if L == 0:
            for t in range(T[0]):
                acc[t] = acc[t] + q[t]
        if kind == 0:
            if L == 0:                                   # input tiles outermost, padded to whole stages of 4 pairs
                for u in range(tin[L]):
                    for t in range(T[L]):
                        acc[t] = _emulate_pair(pairs[k], acts[u], acc[t], last)
                        k += 1
                k = (k + 3) // 4 * 4
            else:
                for t in range(T[L]):
                    for u in range(tin[L]):
                        acc[t] = _emulate_pair(pairs[k], acts[u], acc[t], last)
                        k += 1
        else:                                            # k counts steps: step k holds pairs 4k .. 4k + 3, one per wave
            for u in range(tin[L]):
                for g in range(T[L] // 4):
                    for w in range(4):
                        acc[4 * g + w] = _emulate_pair(pairs[4 * k + w], acts[u], acc[4 * g + w], last)
                    k += 1
        if last:
            last_acc = acc
        else:
            acts = [np.maximum(acc[t], 0.0) for t in range(T[L])]
    assert (k if kind == 0 else 4 * k) == main
    got = _unswap(last_acc, widths[-1], lambda ch: _b_at(bl[n - 1], ch))
    assert np.allclose(got, _want(x, ws, bs), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("cin,widths,xyz_first", [(259, (256, 256, 512), True), (40, (130, 200, 400), False),
                                                   (259, (256, 512, 1024), True), (8, (200, 400, 900), False)])
def test_coop_sa_pack_matches_emulated_dataflow(cin, widths, xyz_first):
    """The cooperative kernel's grouped stacks (coop_mlp.hip): layers in per-wave consumption order (input tiles
    outermost, pair 4k + w belongs to wave w, wave w owns output tiles 4g + w), layer-1 channels [features, xyz].
    Stacks whose last layer is wider than 512 keep layers 1-2 in that order and store the last layer for
    pool_gemm_kernel: [block of four output tiles][contraction tile][tile of the block], bias in plain channel order."""
    from pointnet2_amd import _C
    lib = _C.lib()
    rng = np.random.default_rng(cin)
    dims = (cin,) + tuple(widths)
    ws = [rng.standard_normal((dims[i], dims[i + 1])).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) for i in range(3)]
    info = (ctypes.c_int * 4)()
    wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
    ns = 40                                                           # not a multiple of 32: only this kernel masks
    assert lib.pn2_sa_mlp3_config(cin, *widths, ns, info, ctypes.byref(wf), ctypes.byref(bf)) == 0 and info[0] == 2
    T = [info[1], info[2], info[3]]
    ti = (cin + 31) // 32
    wp = np.empty(wf.value, np.float32)
    bp = np.empty(bf.value, np.float32)
    assert lib.pn2_sa_mlp3_pack(cin, *widths, ns, 1 if xyz_first else 0,
                                *[a.ctypes.data for pair in zip(ws, bs) for a in pair], wp.ctypes.data, bp.ctypes.data) == 0
    pairs = wp.reshape(-1, _PAIR)
    assert pairs.shape[0] == ti * T[0] + T[0] * T[1] + T[1] * T[2]
    ob = np.cumsum([0, T[0] * 32, T[1] * 32, T[2] * 32])
    bl = [bp[ob[i]:ob[i + 1]] for i in range(3)]
    cfeat = cin - 3
    xyz, feat = rng.standard_normal((32, 3)), rng.standard_normal((32, cfeat))
    user_in = np.concatenate([xyz, feat], axis=1) if xyz_first else np.concatenate([feat, xyz], axis=1)
    acts = _operand_tiles(np.concatenate([feat, xyz], axis=1), ti)   # the kernel's channel order
    gemm_last = widths[2] > 512
    tin = [ti, T[0], T[1]]
    lane_bias = lambda L, t: np.stack([bl[L].reshape(-1, 2, 16)[t, l >> 5] for l in range(64)]).astype(np.float64)
    k = 0                                                # step counter: step k holds pairs 4k .. 4k + 3, one per wave
    last_acc = {}
    for L in range(3):
        last = L == 2
        if last and gemm_last:
            break
        acc = {t: (np.zeros((64, 16)) if last else lane_bias(L, t)) for t in range(T[L])}
        for u in range(tin[L]):
            for g in range(T[L] // 4):
                for w in range(4):
                    acc[4 * g + w] = _emulate_pair(pairs[4 * k + w], acts[u], acc[4 * g + w], last)
                k += 1
        if last:
            last_acc = acc
        else:
            acts = [np.maximum(acc[t], 0.0) for t in range(T[L])]
    if gemm_last:
        base = 4 * k
        for t in range(T[2]):                            # pool_gemm_kernel: block cb = t / 4, tile ct = t % 4 of the block
            acc = np.zeros((64, 16))
            for u in range(T[1]):
                acc = _emulate_pair(pairs[base + ((t // 4) * T[1] + u) * 4 + t % 4], acts[u], acc, True)
            last_acc[t] = acc
        got = _unswap(last_acc, widths[2], lambda ch: bl[2][ch])     # plain channel order
    else:
        assert 4 * k == pairs.shape[0]
        got = _unswap(last_acc, widths[2], lambda ch: _b_at(bl[2], ch))
    assert np.allclose(got, _want(user_in, ws, bs), rtol=1e-9, atol=1e-9)


def test_prepare_fused_warms_every_module_of_the_reference_configs():
    """`prepare_fused(device)` then HIP-graph capture must not hit a cold pack cache -- every SA, SA-MSG,
    group_all and FP module of the four reference topologies gets its weights packed ahead of time (host-only work)."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import model_forward_bench as MB
    import pointnet2_amd.pointnet_util as U
    dev = torch.device("cpu")
    # (model, points seen by its group_all level, [(fp module name, c2, c1, unknown points)])
    cases = [(MB.ClsSSG(), 128, []), (MB.ClsMSG(), 128, []),
             (MB.PartSeg(), 128, [("fp1", 1024, 256, 16 * 128), ("fp2", 256, 128, 16 * 512), ("fp3", 128, 6, 16 * 2048)]),
             (MB.SemSeg(), None, [("fp1", 512, 256, 8 * 64), ("fp2", 256, 128, 8 * 256), ("fp3", 256, 64, 8 * 1024),
                                   ("fp4", 128, 0, 8 * 8192)])]
    for model, n_all, fps in cases:
        model.eval()
        for name, mod in model.named_modules():
            if isinstance(mod, U.PointnetSAModule):
                mod.prepare_fused(dev, n=n_all if mod.group_all else None)
                assert mod._pack_cache is not None and len(mod._pack_cache[1]) == 1, name
            elif isinstance(mod, U.PointnetSAModuleMSG):
                mod.prepare_fused(dev)
                assert len(mod._pack_cache) == len(mod.mlps), name
        for name, c2, c1, npts in fps:
            mod = getattr(model, name)
            mod.prepare_fused(c2, c1, npts, dev)
            if name != "fp3" or c1 != 6:                       # part_seg FP3: 134 input channels, no fused kernel covers it
                assert mod._pack_cache is not None and len(mod._pack_cache[1]) == 1, name
