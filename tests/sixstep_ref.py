"""CPU restatement (numpy index maps + the C oracle) of the local steps of hodor_amd/sixstep.py — test
infrastructure shared by the gloo schedule tests (test_sixstep_cpu.py) and the GPU parity tests
(test_gpu_sixstep.py)."""
import numpy as np
import torch

from oracle import pyref as P


class OracleBackend:
    """The local steps of hodor_amd/sixstep.py restated with numpy index maps + the CPU oracle — what
    hodor_sixstep_columns_dev / _rows_dev / _pack_dev / hodor_transpose_dev must compute (the GPU tests
    compare the HIP kernels with this class element for element)."""

    def __init__(self):
        from oracle.oracle import Oracle
        self.O = Oracle(P.BN256.p, P.BN256.g)

    def _np(self, t):
        return t.numpy().view(np.uint64)

    def _fft_rows(self, arr, log_len, omega):
        L = 1 << log_len
        for b in range(len(arr) // L):
            row = np.ascontiguousarray(arr[b * L:(b + 1) * L])
            self.O.serial_fft(row, omega, log_len)
            arr[b * L:(b + 1) * L] = row

    def _twiddle(self, arr, rows, cols, col0, w):
        """arr[r][c] *= w^(r * (col0 + c))"""
        from oracle.oracle import array_to_ints, ints_to_array
        vals = array_to_ints(arr)
        for r in range(rows):
            for c in range(cols):
                vals[r * cols + c] = self.O.mul(vals[r * cols + c], self.O.pow(w, r * (col0 + c)))
        arr[:] = ints_to_array(vals)

    def columns(self, src, log_n1, log_n2, log_p, rank, omega, inverse=False, log_chunks=0, chunk=0, out=None):
        O = self.O
        N1, c2 = 1 << log_n1, 1 << (log_n2 - log_p)
        K, Pn, r1 = 1 << log_chunks, 1 << log_p, 1 << (log_n1 - log_p)
        w = O.inverse(omega) if inverse else omega
        a = self._np(src).copy()                                       # [N1][c2]
        if log_chunks and not inverse:                                 # column group `chunk` of A
            cw = c2 // K
            sub = np.ascontiguousarray(a.reshape(N1, c2, 4)[:, chunk * cw:(chunk + 1) * cw]).reshape(-1, 4)
            res = self._columns_plain(sub, log_n1, log_n2, cw, rank * c2 + chunk * cw, w, False)
            return self._ret(res, out)
        if log_chunks and inverse:                                     # [K][P][rb][c2] received -> [N1][c2]
            rb = r1 // K
            a = np.ascontiguousarray(a.reshape(K, Pn, rb, c2, 4).transpose(1, 0, 2, 3, 4)).reshape(-1, 4)
        return self._ret(self._columns_plain(a, log_n1, log_n2, c2, rank * c2, w, inverse), out)

    def _ret(self, arr, out):
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64))
        if out is not None:
            out.copy_(t)
            return out
        return t

    def _columns_plain(self, a, log_n1, log_n2, c2, col0, w, inverse):
        O = self.O
        N1 = 1 << log_n1
        if inverse:
            self._twiddle(a, N1, c2, col0, w)                          # w^-(k1 * n2) on the way in
        t = np.ascontiguousarray(a.reshape(N1, c2, 4).transpose(1, 0, 2)).reshape(c2 * N1, 4)
        self._fft_rows(t, log_n1, O.pow(w, 1 << log_n2))
        a = np.ascontiguousarray(t.reshape(c2, N1, 4).transpose(1, 0, 2)).reshape(N1 * c2, 4)
        if not inverse:
            self._twiddle(a, N1, c2, col0, w)                          # w^(k1 * n2) on the way out
        else:
            ninv = np.array([[(O.inverse(O.from_canonical(1 << (log_n1 + log_n2))) >> (64 * i)) & (2**64 - 1)
                              for i in range(4)]], dtype=np.uint64)
            b = np.repeat(ninv, len(a), axis=0)
            O.poly_binary(a, b, "mul")
        return a

    def rows(self, src, log_n1, log_n2, log_p, rank, omega, inverse=False, log_chunks=0, chunk=0, out=None):
        O = self.O
        Pn, r1, c2 = 1 << log_p, 1 << (log_n1 - log_p), 1 << (log_n2 - log_p)
        K = 1 << log_chunks
        w = O.inverse(omega) if inverse else omega
        a = self._np(src).copy()
        if not inverse:                                      # [K][P][r1][cw] as received -> [r1][N2] (n2 = s*c2 + k*cw + j)
            cw = c2 // K
            a = np.ascontiguousarray(a.reshape(K, Pn, r1, cw, 4).transpose(2, 1, 0, 3, 4)).reshape(-1, 4)
        else:                                                # row group `chunk` of B
            rb = r1 // K
            a = np.ascontiguousarray(a.reshape(r1, -1, 4)[chunk * rb:(chunk + 1) * rb]).reshape(-1, 4)
            r1 = rb
        self._fft_rows(a, log_n2, O.pow(w, 1 << log_n1))
        if inverse:                                          # [rb][N2] -> the slabs to send, [P][rb][c2]
            a = np.ascontiguousarray(a.reshape(r1, Pn, c2, 4).transpose(1, 0, 2, 3)).reshape(-1, 4)
        return self._ret(a, out)

    def pack(self, src, log_rows, log_cols, log_p):
        rows, Pn, c = 1 << log_rows, 1 << log_p, 1 << (log_cols - log_p)
        a = self._np(src).reshape(rows, Pn, c, 4).transpose(1, 0, 2, 3)
        return torch.from_numpy(np.ascontiguousarray(a).reshape(-1, 4).view(np.int64))

    def transpose(self, src, rows, cols):
        a = self._np(src).reshape(rows, cols, 4).transpose(1, 0, 2)
        return torch.from_numpy(np.ascontiguousarray(a).reshape(-1, 4).view(np.int64))

    def scale(self, buf, s):
        arr = np.ascontiguousarray(self._np(buf))
        self.O.poly_unary(arr, "scale", c=s)
        self._np(buf)[:] = arr
        return buf

    def batched_ntt(self, buf, batch, log_len, omega):
        out = buf.clone()
        self._fft_rows(self._np(out), log_len, omega)
        return out

    def distribute_powers(self, buf, g):
        arr = np.ascontiguousarray(self._np(buf))
        self.O.distribute_powers(arr, g)
        self._np(buf)[:] = arr
        return buf

    def pow(self, a, e):
        return self.O.pow(a, e)

    def mul(self, a, b):
        return self.O.mul(a, b)

    def inverse(self, a):
        return self.O.inverse(a)

    def from_u64(self, v):
        return self.O.from_canonical(v)


def layout_a(full, log_n, rank, world):
    """Column block `rank` of the N1 x N2 matrix x[n1*N2 + n2] (hodor_amd/sixstep.py layout A)."""
    from hodor_amd.sixstep import split_logs
    log_n1, log_n2 = split_logs(log_n)
    c2 = (1 << log_n2) // world
    m = full.reshape(1 << log_n1, 1 << log_n2, 4)
    return np.ascontiguousarray(m[:, rank * c2:(rank + 1) * c2]).reshape(-1, 4)


def layout_b(spectrum, log_n, rank, world):
    """Row block `rank` of the N1 x N2 matrix X[k1 + N1*k2] (layout B)."""
    from hodor_amd.sixstep import split_logs
    log_n1, log_n2 = split_logs(log_n)
    r1 = (1 << log_n1) // world
    m = spectrum.reshape(1 << log_n2, 1 << log_n1, 4).transpose(1, 0, 2)      # [k1][k2]
    return np.ascontiguousarray(m[rank * r1:(rank + 1) * r1]).reshape(-1, 4)


def layout_a_torch(full, log_n1, log_n2, rank, world):
    """layout_a on a device tensor (n, 4): column block `rank` of the N1 x N2 matrix, contiguous."""
    c2 = (1 << log_n2) // world
    return full.view(1 << log_n1, 1 << log_n2, 4)[:, rank * c2:(rank + 1) * c2].contiguous().view(-1, 4)


def layout_b_torch(spectrum, log_n1, log_n2, rank, world):
    """layout_b on a device tensor: row block `rank` of the N1 x N2 matrix X[k1 + N1*k2]."""
    r1 = (1 << log_n1) // world
    m = spectrum.view(1 << log_n2, 1 << log_n1, 4).permute(1, 0, 2)            # [k1][k2]
    return m[rank * r1:(rank + 1) * r1].contiguous().view(-1, 4)
