"""Host-side packing of the fused SA MLP (pn2_sa_mlp3_pack) against a numpy emulation of the kernel's
MFMA data flow (CPU; no device work). The emulation hard-codes the documented operand maps of
v_mfma_f32_32x32x2_f32 -- A: lane l holds A[i = l & 31][k = l >> 5]; B: B[k = l >> 5][j = l & 31];
C/D: lane l, register v holds D[8(v >> 2) + 4(l >> 5) + (v & 3)][l & 31] -- so it proves that the
permutation algebra (weights, bias, register-to-channel map, layer chaining) is right GIVEN those maps;
the maps themselves are checked on the GPU by tests/test_sa_mlp_gpu.py."""
import ctypes

import numpy as np
import pytest


def _chan(v, h):
    return 8 * (v >> 2) + 4 * h + (v & 3)


def _mfma(a, b, acc):
    """One v_mfma_f32_32x32x2_f32: a, b (64,) lane operands, acc (64, 16) -> acc + A.B in the C/D map."""
    A = np.zeros((32, 2))
    B = np.zeros((2, 32))
    for l in range(64):
        A[l & 31, l >> 5] = a[l]
        B[l >> 5, l & 31] = b[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for v in range(16):
            out[l, v] += D[8 * (v >> 2) + 4 * (l >> 5) + (v & 3), l & 31]
    return out


def _emulate_layer(wp, bp, t_out, t_in, acts, last=False):
    """acts: list of t_in arrays (64 lanes, 16 regs) -> list of t_out arrays, exactly as mlp_layer walks them.
    last: the kernel swaps the MFMA operands and starts from zero (bias + ReLU come after the pooling)."""
    wp = wp.reshape(t_out, t_in, 4, 64, 4)
    bp = bp.reshape(t_out, 2, 16)
    outs = []
    for t in range(t_out):
        acc = np.zeros((64, 16)) if last else np.stack([bp[t, l >> 5] for l in range(64)]).astype(np.float64)
        for u in range(t_in):
            for q in range(4):
                for r in range(4):
                    w, x = wp[t, u, q, :, r], acts[u][:, 4 * q + r]
                    acc = _mfma(x, w, acc) if last else _mfma(w, x, acc)
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
    tiles = (ctypes.c_int * 3)()
    wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
    assert lib.pn2_sa_mlp3_config(cin, *widths, tiles, ctypes.byref(wf), ctypes.byref(bf)) == 0
    t1, t2, t3 = tiles
    wp = np.empty(wf.value, np.float32)
    bp = np.empty(bf.value, np.float32)
    assert lib.pn2_sa_mlp3_pack(cin, *widths, *[a.ctypes.data for pair in zip(ws, bs) for a in pair], wp.ctypes.data,
                                bp.ctypes.data) == 0
    x = rng.standard_normal((32, cin)).astype(np.float32)          # 32 samples
    # layer-1 operand registers: register v of lane l = input channel chan(v, l >> 5) of sample l & 31
    x0 = np.zeros((64, 16))
    for l in range(64):
        for v in range(16):
            k = _chan(v, l >> 5)
            x0[l, v] = x[l & 31, k] if k < cin else 0.0
    sizes_w = [t1 * 1 * 1024, t2 * t1 * 1024, t3 * t2 * 1024]
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
    want = x.astype(np.float64)
    for w, b in zip(ws, bs):
        want = np.maximum(want @ w.astype(np.float64) + b, 0.0)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)


def test_config_limits():
    from pointnet2_amd import _C
    lib = _C.lib()
    assert lib.pn2_sa_mlp3_config(3, 64, 64, 128, None, None, None) == 0
    assert lib.pn2_sa_mlp3_config(35, 64, 64, 128, None, None, None) != 0       # too many input channels
    assert lib.pn2_sa_mlp3_config(3, 128, 128, 256, None, None, None) != 0      # SA2-sized: unfused path
    assert lib.pn2_sa_mlp3_config(2, 64, 64, 128, None, None, None) != 0
