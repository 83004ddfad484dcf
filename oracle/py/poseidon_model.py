"""ORACLE -- test infrastructure only.  Big-int model of Poseidon over Goldilocks (width 12, x^7, 8 + 22 rounds) and of the Merkle-cap
commitment, written independently of oracle/c/tmxo_poseidon.c (Python integers, explicit matrix form of the MDS layer, the inverse
permutation for the bijection self-check).  Constants: see the header of tmxo_poseidon.c -- the defaults are the Poseidon paper's Grain-LFSR
stream for these parameters, NOT plonky2's table; parity unpinned."""
P = 2**64 - 2**32 + 1
T, RF, RP = 12, 8, 22
MDS_CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
MDS_DIAG = [8] + [0] * 11


def grain_constants(count=T * (RF + RP), field=1, sbox=0, n=64, t=T, rf=RF, rp=RP, p=P):
    bits = []
    for v, w in ((field, 2), (sbox, 4), (n, 12), (t, 12), (rf, 10), (rp, 10)):
        bits += [(v >> i) & 1 for i in range(w - 1, -1, -1)]
    st = bits + [1] * 30

    def nxt():
        b = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0]
        st.pop(0)
        st.append(b)
        return b
    for _ in range(160):
        nxt()
    out = []
    while len(out) < count:
        v, got = 0, 0
        while got < n:
            a, c = nxt(), nxt()
            if a:
                v, got = (v << 1) | c, got + 1
        if v < p:
            out.append(v)
    return out


def mds_matrix(circ=MDS_CIRC, diag=MDS_DIAG):
    return [[(circ[(c - r) % T] + (diag[r] if c == r else 0)) % P for c in range(T)] for r in range(T)]


def mat_vec(m, v):
    return [sum(m[r][c] * v[c] for c in range(T)) % P for r in range(T)]


def mat_inverse(m):
    n = len(m)
    a = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(m)]
    for col in range(n):
        piv = next(r for r in range(col, n) if a[r][col] % P)
        a[col], a[piv] = a[piv], a[col]
        inv = pow(a[col][col], P - 2, P)
        a[col] = [x * inv % P for x in a[col]]
        for r in range(n):
            if r != col and a[r][col]:
                f = a[r][col]
                a[r] = [(x - f * y) % P for x, y in zip(a[r], a[col])]
    return [row[n:] for row in a]


class Poseidon:
    def __init__(self, rc=None, circ=None, diag=None):
        self.rc = [x % P for x in (rc or grain_constants())]
        self.m = mds_matrix(circ or MDS_CIRC, diag or MDS_DIAG)
        self.m_inv = None

    def permute(self, s):
        s = [x % P for x in s]
        for r in range(RF + RP):
            s = [(x + c) % P for x, c in zip(s, self.rc[r * T:(r + 1) * T])]
            if r < RF // 2 or r >= RF // 2 + RP:
                s = [pow(x, 7, P) for x in s]
            else:
                s[0] = pow(s[0], 7, P)
            s = mat_vec(self.m, s)
        return s

    def permute_inverse(self, s):
        """undo permute(): MDS^-1, S-box^-1 (x -> x^d, 7 d = 1 mod p - 1), subtract the constants, last round first"""
        if self.m_inv is None:
            self.m_inv = mat_inverse(self.m)
        d = pow(7, -1, P - 1)
        s = [x % P for x in s]
        for r in range(RF + RP - 1, -1, -1):
            s = mat_vec(self.m_inv, s)
            if r < RF // 2 or r >= RF // 2 + RP:
                s = [pow(x, d, P) for x in s]
            else:
                s[0] = pow(s[0], d, P)
            s = [(x - c) % P for x, c in zip(s, self.rc[r * T:(r + 1) * T])]
        return s

    def hash_no_pad(self, xs):
        s = [0] * T
        for off in range(0, len(xs), 8):
            chunk = xs[off:off + 8]
            s[:len(chunk)] = [x % P for x in chunk]
            s = self.permute(s)
        return s[:4]

    def two_to_one(self, l, r):
        return self.permute(list(l) + list(r) + [0] * 4)[:4]

    def merkle(self, rows, cap_height):
        """rows: list of rows (each a list of column values).  Returns the list of levels, leaves first, the cap last."""
        width = len(rows[0])
        level = [([x % P for x in r] + [0] * 4)[:4] if width <= 4 else self.hash_no_pad(r) for r in rows]
        levels = [level]
        while len(level) > (1 << cap_height):
            level = [self.two_to_one(level[2 * i], level[2 * i + 1]) for i in range(len(level) // 2)]
            levels.append(level)
        return levels
