#!/usr/bin/env python3
"""Generates tests/golden/stark_kat.json with an INDEPENDENT big-integer Python
restatement of this repository's proof-system primitives: BabyBear,
the quartic extension, Poseidon2 (constants from hashlib's SHAKE256), sponge /
compression, the duplex challenger, an O(n^2) DFT and one FRI fold.

Neither the C oracle nor the HIP code is imported here: the JSON pins both to a
third implementation.  No reference vectors exist for these primitives
("no golden proof bytes ... are committed anywhere"), so
parity with SP1/Plonky3 stays unpinned; this file pins self-consistency.

Run:  python tests/golden/gen_golden.py   (rewrites stark_kat.json next to it)
"""
import hashlib
import json
import os

P = 2013265921
G = 31
W = 11


def constants():
    stream = hashlib.shake_256(b"zksp/poseidon2/babybear/w16/v1").digest(136 * 16)
    vals, pos = [], 0
    while len(vals) < 141:
        w = int.from_bytes(stream[pos:pos + 4], "little") & 0x7FFFFFFF
        pos += 4
        if w < P:
            vals.append(w)
    ext = [vals[16 * r:16 * r + 16] for r in range(4)]
    internal = vals[64:77]
    ext += [vals[77 + 16 * r:77 + 16 * r + 16] for r in range(4)]
    diag = [P - 2] + [1 << (i - 1) for i in range(1, 15)] + [1 << 15]
    return ext, internal, diag


EXT, INT, DIAG = constants()
M4 = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]


def ext_linear(s):
    out = []
    for c in range(4):
        v = s[4 * c:4 * c + 4]
        out += [sum(M4[i][j] * v[j] for j in range(4)) % P for i in range(4)]
    sums = [(out[j] + out[4 + j] + out[8 + j] + out[12 + j]) % P for j in range(4)]
    return [(out[i] + sums[i % 4]) % P for i in range(16)]


def int_linear(s):
    t = sum(s) % P
    return [(s[i] * DIAG[i] + t) % P for i in range(16)]


def permute(s):
    s = ext_linear(list(s))
    for r in range(4):
        s = ext_linear([pow((s[i] + EXT[r][i]) % P, 7, P) for i in range(16)])
    for r in range(13):
        s[0] = pow((s[0] + INT[r]) % P, 7, P)
        s = int_linear(s)
    for r in range(4, 8):
        s = ext_linear([pow((s[i] + EXT[r][i]) % P, 7, P) for i in range(16)])
    return s


def hash_elems(v):
    st = [0] * 16
    for off in range(0, len(v), 8):
        chunk = list(v[off:off + 8])
        st[:8] = chunk + [0] * (8 - len(chunk))  # overwrite mode; the last block is zero-filled
        st = permute(st)
    return st[:8]


def compress(l, r):
    return permute(list(l) + list(r))[:8]


class Challenger:
    def __init__(self):
        self.state, self.inbuf, self.outbuf = [0] * 16, [], []

    def duplex(self):
        self.state[:len(self.inbuf)] = self.inbuf
        self.inbuf = []
        self.state = permute(self.state)
        self.outbuf = self.state[:8]

    def observe(self, x):
        self.outbuf = []
        self.inbuf.append(x)
        if len(self.inbuf) == 8:
            self.duplex()

    def sample(self):
        if self.inbuf or not self.outbuf:
            self.duplex()
        return self.outbuf.pop()


def e_mul(a, b):
    t = [0] * 7
    for i in range(4):
        for j in range(4):
            t[i + j] += a[i] * b[j]
    return [(t[i] + W * t[i + 4]) % P if i < 3 else t[i] % P for i in range(4)]


def e_pow(a, e):
    r = [1, 0, 0, 0]
    while e:
        if e & 1:
            r = e_mul(r, a)
        a = e_mul(a, a)
        e >>= 1
    return r


def e_inv(a):
    return e_pow(a, P ** 4 - 2)


def root_of_unity(logn):
    return pow(G, (P - 1) >> logn, P)


def dft(v):
    n = len(v)
    w = root_of_unity(n.bit_length() - 1)
    return [sum(v[k] * pow(w, i * k, P) for k in range(n)) % P for i in range(n)]


def fri_fold(layer, shift, beta):
    """layer[c][m] ext, coset-major; returns the folded layer."""
    hk = len(layer[0])
    half = hk // 2
    loghk = hk.bit_length() - 1
    w2, w = root_of_unity(loghk + 1), root_of_unity(loghk)
    inv2 = pow(2, -1, P)
    out = []
    for c in range(2):
        base = shift * (w2 if c else 1) % P
        row = []
        for m in range(half):
            lo, hi = layer[c][m], layer[c][m + half]
            x = base * pow(w, m, P) % P
            s = [(lo[i] + hi[i]) * inv2 % P for i in range(4)]
            d = [(lo[i] - hi[i]) * inv2 * pow(x, -1, P) % P for i in range(4)]
            bd = e_mul(beta, d)
            row.append([(s[i] + bd[i]) % P for i in range(4)])
        out.append(row)
    return out


def lcg(seed):
    x = seed
    while True:
        x = (x * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        yield (x >> 33) % P


def main():
    rnd = lcg(2024)
    vec = lambda n: [next(rnd) for _ in range(n)]
    out = {"p": P, "ext_rc": EXT, "int_rc": INT, "int_diag": DIAG}
    states = [[0] * 16, list(range(16)), [P - 1] * 16, vec(16)]
    out["permute"] = [{"in": s, "out": permute(s)} for s in states]
    out["hash_elems"] = [{"in": v, "out": hash_elems(v)} for v in ([], [5], vec(8), vec(9), vec(37))]
    l, r = vec(8), vec(8)
    out["compress"] = {"l": l, "r": r, "out": compress(l, r)}
    ch = Challenger()
    obs1, obs2 = vec(11), vec(3)
    for x in obs1:
        ch.observe(x)
    s1 = [ch.sample() for _ in range(5)]
    for x in obs2:
        ch.observe(x)
    s2 = [ch.sample() for _ in range(9)]
    out["challenger"] = {"observe1": obs1, "sample1": s1, "observe2": obs2, "sample2": s2}
    a, b = vec(4), vec(4)
    out["ext"] = {"a": a, "b": b, "mul": e_mul(a, b), "inv_a": e_inv(a)}
    v = vec(16)
    out["dft"] = {"in": v, "out": dft(v)}
    layer = [[vec(4) for _ in range(8)] for _ in range(2)]
    beta = vec(4)
    out["fri_fold"] = {"layer": layer, "shift": 31, "beta": beta, "out": fri_fold(layer, 31, beta)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stark_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
