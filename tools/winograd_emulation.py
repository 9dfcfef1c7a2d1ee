#!/usr/bin/env python3
"""CPU emulation (numpy) of a 3x3x3 convolution, 64 -> 64 channels, under the arithmetic of csrc/conv_bf.hip (operands
range-scaled by a power of two, split into fp16 hi + lo, three products per block of 16 channels, fp32 accumulation)
computed (a) directly (27 taps), (b) by Winograd F(2x2x2, 3x3x3): input / filter / output transforms in fp32, the
TRANSFORMED operands split into hi + lo (64 instead of 216 products per 8 outputs), (c) by the 1-D F(2, 3) along x
only (2 instead of 3 products per output), against the fp64 convolution.

Verdict r5 item 3: adopt only if max |error| <= 2e-6 of the tensor maximum (and >= 1.4x faster on the GPU).
Run: python tools/winograd_emulation.py  [T = tiles per axis, default 6 -> 12^3 outputs]
"""
import sys
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)       # 4 x 4
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)                  # 4 x 3
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)                                      # 2 x 4


def range_scale(x):
    m = float(np.abs(x).max())
    return np.float32(2.0 ** (15 - np.frexp(m)[1])) if m > 0 else np.float32(1.0)


def split_f16(x, scale):
    r = (x.astype(np.float32) * scale).astype(np.float32)
    hi = r.astype(np.float16).astype(np.float32)
    lo = (r - hi).astype(np.float32).astype(np.float16).astype(np.float32)
    return hi, lo


def gemm_f16x3(A, B):
    """(M, K) x (K, N) with both operands split; K in blocks of 16, block sums exact, fp32 accumulator (the MFMA model)."""
    sa, sb = range_scale(A), range_scale(B)
    ah, al = split_f16(A, sa)
    bh, bl = split_f16(B, sb)
    acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k in range(0, A.shape[1], 16):
        s = slice(k, k + 16)
        part = (ah[:, s].astype(np.float64) @ bh[s].astype(np.float64) + ah[:, s].astype(np.float64) @ bl[s].astype(np.float64)
                + al[:, s].astype(np.float64) @ bh[s].astype(np.float64))
        acc = (acc.astype(np.float64) + part).astype(np.float32)
    return (acc / (sa * sb)).astype(np.float32)


def mode_mul(M, x, axis, dtype):
    """apply matrix M along `axis` of x in `dtype` arithmetic (sequential adds, as a kernel would)."""
    x = np.moveaxis(x, axis, 0).astype(dtype)
    out = np.zeros((M.shape[0],) + x.shape[1:], dtype)
    for i in range(M.shape[0]):
        for j in range(M.shape[1]):
            if M[i, j] != 0:
                out[i] = (out[i] + dtype(M[i, j]) * x[j]).astype(dtype)
    return np.moveaxis(out, 0, axis)


def conv_ref(x, w):
    """x (D+2, H+2, W+2, Cin) already padded, w (3, 3, 3, Cin, Cout) -> (D, H, W, Cout), fp64."""
    D, H, W = (s - 2 for s in x.shape[:3])
    y = np.zeros((D, H, W, w.shape[-1]), np.float64)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                y += x[a:a + D, b:b + H, c:c + W].astype(np.float64) @ w[a, b, c].astype(np.float64)
    return y


def conv_direct_f16x3(x, w):
    D, H, W = (s - 2 for s in x.shape[:3])
    Cin = x.shape[-1]
    cols = np.concatenate([x[a:a + D, b:b + H, c:c + W].reshape(-1, Cin) for a in range(3) for b in range(3) for c in range(3)], 1)
    return gemm_f16x3(cols, w.reshape(27 * Cin, -1)).reshape(D, H, W, -1)


def conv_winograd3d(x, w, split=True):
    D, H, W = (s - 2 for s in x.shape[:3])
    Cin, Cout = w.shape[3:]
    T = (D // 2, H // 2, W // 2)
    # filter transform (once per weight update; fp64 then fp32): U (4,4,4,Cin,Cout)
    U = w.astype(np.float64)
    for ax in range(3):
        U = mode_mul(G, U, ax, np.float64)
    U = U.astype(np.float32)
    # input tiles (T0,T1,T2,4,4,4,Cin), transform in fp32
    idx = lambda n: (2 * np.arange(n))[:, None] + np.arange(4)[None]
    d = x[idx(T[0])[:, None, None, :, None, None], idx(T[1])[None, :, None, None, :, None], idx(T[2])[None, None, :, None, None, :]]
    V = d.astype(np.float32)
    for ax in (3, 4, 5):
        V = mode_mul(BT, V, ax, np.float32)
    V = V.reshape(-1, 64, Cin)
    Uf = U.reshape(64, Cin, Cout)
    M = np.empty((V.shape[0], 64, Cout), np.float32)
    for p in range(64):
        M[:, p] = gemm_f16x3(V[:, p], Uf[p]) if split else (V[:, p].astype(np.float64) @ Uf[p].astype(np.float64)).astype(np.float32)
    M = M.reshape(T + (4, 4, 4, Cout))
    for ax in (3, 4, 5):
        M = mode_mul(AT, M, ax, np.float32)
    return M.transpose(0, 3, 1, 4, 2, 5, 6).reshape(D, H, W, Cout)


def conv_winograd1d(x, w):
    """F(2, 3) along x only: 9 (z, y) taps x 4 transformed x positions per 2 outputs."""
    D, H, W = (s - 2 for s in x.shape[:3])
    Cin, Cout = w.shape[3:]
    U = mode_mul(G, w.astype(np.float64), 2, np.float64).astype(np.float32)                   # (3,3,4,Cin,Cout)
    idx = (2 * np.arange(W // 2))[:, None] + np.arange(4)[None]
    V = mode_mul(BT, x[:, :, idx].astype(np.float32), 3, np.float32)                           # (D+2,H+2,W/2,4,Cin)
    M = np.empty((D * H * (W // 2), 4, Cout), np.float32)
    for p in range(4):
        cols = np.concatenate([V[a:a + D, b:b + H, :, p].reshape(-1, Cin) for a in range(3) for b in range(3)], 1)
        M[:, p] = gemm_f16x3(cols, U[:, :, p].reshape(9 * Cin, Cout))
    y = mode_mul(AT, M.reshape(D, H, W // 2, 4, Cout), 3, np.float32)
    return y.reshape(D, H, W, Cout)


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.default_rng(0)
    Cin = Cout = 64
    S = 2 * T
    cases = {
        "GroupNorm+ReLU activations x He-uniform weights (forward)":
            (np.maximum(rng.standard_normal((S, S, S, Cin)) * 1.2 + 0.1, 0), rng.uniform(-1, 1, (3, 3, 3, Cin, Cout)) / np.sqrt(27 * Cin)),
        "smooth field + noise (image-like) x weights":
            (np.maximum(np.cumsum(rng.standard_normal((S, S, S, Cin)), 2) * 0.3 + rng.standard_normal((S, S, S, Cin)) * 0.1, 0),
             rng.uniform(-1, 1, (3, 3, 3, Cin, Cout)) / np.sqrt(27 * Cin)),
        "gradients (log-normal magnitudes ~1e-5) x weights (data gradient)":
            (rng.standard_normal((S, S, S, Cin)) * np.exp(rng.standard_normal((S, S, S, Cin))) * 1e-5,
             rng.uniform(-1, 1, (3, 3, 3, Cin, Cout)) / np.sqrt(27 * Cin)),
    }
    print(f"64 -> 64, {S}^3 outputs; error = max |y - fp64| / max |fp64|   (bar: 2e-6)")
    for name, (x, w) in cases.items():
        x = np.pad(x.astype(np.float32), ((1, 1), (1, 1), (1, 1), (0, 0)))
        w = w.astype(np.float32)
        ref = conv_ref(x, w)
        mx = np.abs(ref).max()
        f32 = np.zeros(ref.shape, np.float32)                # plain sequential fp32 (what the reference's CPU path is at best)
        for a in range(3):
            for b in range(3):
                for c in range(3):
                    xs = x[a:a + S, b:b + S, c:c + S]
                    for k in range(0, Cin, 8):
                        f32 = (f32 + (xs[..., k:k + 8] @ w[a, b, c, k:k + 8]).astype(np.float32)).astype(np.float32)
        res = {"fp32 direct": f32, "f16x3 direct (HEAD)": conv_direct_f16x3(x, w),
               "f16x3 Winograd F(2,3) along x": conv_winograd1d(x, w),
               "f16x3 Winograd F(2x2x2,3x3x3)": conv_winograd3d(x, w),
               "exact-product Winograd F(2x2x2,3x3x3) (fp32 transforms only)": conv_winograd3d(x, w, split=False)}
        print(name)
        for k, y in res.items():
            e = np.abs(y - ref)
            print(f"   {k:62s} max {e.max() / mx:.2e}   mean {e.mean() / mx:.2e}")
