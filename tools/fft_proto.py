"""Prototype (numpy, float64) of the in-place LDS convolution of csrc/fftconv.h: validates the index algebra.
   1-D: L = R1*R2, step A (radix R1 over stride R2, twiddle w_L^(n2 k1)), step B (radix R2, contiguous);
   X[k1 + R1 k2] ends at position R2 k1 + k2.  Real rows by the half-length trick, pairs (k, M-k) in place."""
import numpy as np
from scipy.signal import convolve2d

def dft(v, inv):  # v: (R, ...) along axis 0
    R = v.shape[0]
    n = np.arange(R)
    Wm = np.exp((2j if inv else -2j) * np.pi * np.outer(n, n) / R)
    return np.tensordot(Wm, v, axes=(1, 0))

def fft_pass(A, lines, line_stride, J, joff, R, qs, tw, inv, use_tw):
    """A flat complex array; in place"""
    for line in range(lines):
        for j in range(J):
            base = line * line_stride + j * joff
            idx = base + np.arange(R) * qs
            v = dft(A[idx], inv)
            if use_tw:
                t = tw[(j * np.arange(R))]
                v = v * (np.conj(t) if inv else t)
            A[idx] = v

def fwd1d(A, lines, line_stride, es, R1, R2, tw):
    fft_pass(A, lines, line_stride, R2, es, R1, R2 * es, tw, False, True)       # A
    fft_pass(A, lines, line_stride, R1, R2 * es, R2, es, tw, False, False)      # B
def inv1d(A, lines, line_stride, es, R1, R2, tw):
    fft_pass(A, lines, line_stride, R1, R2 * es, R2, es, tw, True, True)        # B^-1
    fft_pass(A, lines, line_stride, R2, es, R1, R2 * es, tw, True, False)       # A^-1

def test_1d():
    rng = np.random.RandomState(0)
    for R1, R2 in ((8, 10), (12, 7), (10, 16), (5, 9), (16, 16), (7, 3)):
        L = R1 * R2
        x = rng.randn(L) + 1j * rng.randn(L)
        A = x.copy(); tw = np.exp(-2j * np.pi * np.arange(L) / L)
        fwd1d(A, 1, L, 1, R1, R2, tw)
        X = np.fft.fft(x)
        p = np.arange(L); k = (p // R2) + R1 * (p % R2)
        assert np.allclose(A, X[k]), (R1, R2)
        inv1d(A, 1, L, 1, R1, R2, tw)
        assert np.allclose(A / L, x)
    print("1d ok")

class Plan:
    def __init__(s, H, W, Fy, Fx, R1y, R2y, R1x, R2x):
        s.H, s.W, s.Fy, s.Fx = H, W, Fy, Fx
        s.M = Fx // 2
        assert Fx % 2 == 0 and R1y * R2y == Fy and R1x * R2x == s.M
        s.R1y, s.R2y, s.R1x, s.R2x = R1y, R2y, R1x, R2x
        s.RS = s.M + 1
        s.twy = np.exp(-2j * np.pi * np.arange(Fy) / Fy)
        s.twm = np.exp(-2j * np.pi * np.arange(s.M) / s.M)
        s.twx = np.exp(-2j * np.pi * np.arange(s.M // 2 + 1) / Fx)
        k = np.arange(s.M)
        s.posx = R2x * (k % R1x) + k // R1x

def rows_fwd(p, A, nrows):
    """A: (Fy, RS) complex, rows < nrows hold z[n] = x[2n] + i x[2n+1] in [0, M)"""
    flat = A.reshape(-1)
    fwd1d(flat, nrows, p.RS, 1, p.R1x, p.R2x, p.twm)
    M = p.M
    for y in range(nrows):
        r = A[y]
        z0 = r[p.posx[0]]
        r[p.posx[0]] = z0.real + z0.imag
        r[M] = z0.real - z0.imag
        for k in range(1, (M + 1) // 2):
            a, b = r[p.posx[k]], r[p.posx[M - k]]
            w = p.twx[k]
            xk = 0.5 * (a + np.conj(b)) - 0.5j * w * (a - np.conj(b))
            xmk = 0.5 * (b + np.conj(a)) + 0.5j * np.conj(w) * (b - np.conj(a))
            r[p.posx[k]], r[p.posx[M - k]] = xk, xmk
        if M % 2 == 0:
            r[p.posx[M // 2]] = np.conj(r[p.posx[M // 2]])

def rows_inv(p, A, nrows):
    M = p.M
    for y in range(nrows):
        r = A[y]
        x0, xM = r[p.posx[0]].real, r[M].real     # both real for a real signal
        # E0 = (X0 + XM)/2, O0 = (X0 - XM)/2 ; Z0 = E0 + i O0
        r[p.posx[0]] = 0.5 * (x0 + xM) + 0.5j * (x0 - xM)
        for k in range(1, (M + 1) // 2):
            xk, xmk = r[p.posx[k]], r[p.posx[M - k]]
            w = p.twx[k]
            E = 0.5 * (xk + np.conj(xmk)); O = 0.5 * (xk - np.conj(xmk)) * np.conj(w)
            zk = E + 1j * O
            # Z[M-k] = E[M-k] + i O[M-k] = conj(E[k]) + i conj(O[k])
            zmk = np.conj(E) + 1j * np.conj(O)
            r[p.posx[k]], r[p.posx[M - k]] = zk, zmk
        if M % 2 == 0:
            r[p.posx[M // 2]] = np.conj(r[p.posx[M // 2]])
    inv1d(A.reshape(-1), nrows, p.RS, 1, p.R1x, p.R2x, p.twm)

def fwd2d(p, A, nrows):
    rows_fwd(p, A, nrows)
    fwd1d(A.reshape(-1), p.M + 1, 1, p.RS, p.R1y, p.R2y, p.twy)
def inv2d(p, A, nrows):
    inv1d(A.reshape(-1), p.M + 1, 1, p.RS, p.R1y, p.R2y, p.twy)
    rows_inv(p, A, nrows)

def load_real(p, img, oy=0, ox=0):
    """real image placed at (y+oy) mod Fy, (x+ox) mod Fx, packed as pairs"""
    full = np.zeros((p.Fy, p.Fx))
    h, w = img.shape
    ys = (np.arange(h) + oy) % p.Fy; xs = (np.arange(w) + ox) % p.Fx
    full[np.ix_(ys, xs)] = img
    A = np.zeros((p.Fy, p.RS), complex)
    A[:, :p.M] = full[:, 0::2] + 1j * full[:, 1::2]
    return A
def unpack_real(p, A):
    full = np.zeros((p.Fy, p.Fx))
    full[:, 0::2] = A[:, :p.M].real; full[:, 1::2] = A[:, :p.M].imag
    return full

def test_conv():
    import sys; sys.path.insert(0, "/root/repo")
    from oracle import pgm
    rng = np.random.RandomState(1)
    for (H, W, P, Q, Fy, Fx, fy, fx) in ((128, 128, 41, 41, 160, 160, (10, 16), (8, 10)), (58, 48, 43, 43, 84, 72, (12, 7), (4, 9)),
                                       (23, 30, 7, 9, 28, 36, (4, 7), (2, 9)), (31, 55, 41, 41, 56, 80, (7, 8), (5, 8)),
                                       (33, 20, 6, 9, 40, 28, (5, 8), (2, 7))):
        p = Plan(H, W, Fy, Fx, fy[0], fy[1], fx[0], fx[1])
        img = rng.rand(H, W); ker = rng.rand(P, Q)
        Fry = pgm.next_fast_len(H + P + 3); Frx = pgm.next_fast_len(W + Q + 3)
        while Frx & 1: Frx = pgm.next_fast_len(Frx + 1)
        oky = (Fry - P + 1) // 2 - Fry // 2; okx = (Frx - Q + 1) // 2 - Frx // 2
        Kh = load_real(p, ker, oky, okx); fwd2d(p, Kh, p.Fy)
        Kh /= (p.M * p.Fy)
        A = load_real(p, img); fwd2d(p, A, H)
        A *= Kh; inv2d(p, A, H)
        out = unpack_real(p, A)[:H, :W]
        ref = pgm.convolve(img[None], ker[None], axes=(1, 2))[0]
        err = np.abs(out - ref).max() / np.abs(ref).max()
        # adjoint: <conv(x), y> == <x, adj(y)>
        yv = rng.rand(H, W)
        B = load_real(p, yv); fwd2d(p, B, H); B *= np.conj(Kh); inv2d(p, B, H)
        adj = unpack_real(p, B)[:H, :W]
        ref_adj = pgm.render_adjoint(yv[None], ker[None])[0]
        err2 = np.abs(adj - ref_adj).max() / np.abs(ref_adj).max()
        print((H, W, P, Q, Fy, Fx), "conv err %.2e adjoint err %.2e  minF %d %d" % (err, err2, H - 1 - oky + 1, 0))
        assert err < 1e-12 and err2 < 1e-12

if __name__ == "__main__":
    test_1d(); test_conv()
